// recon_kernel.inl -- k_recon: dequantise + classify + 8x8 IDCT + half-pel motion
// compensation + residual add + clip, one workgroup per tile of 8x2 macroblocks.
//
// Replaces, per picture: the numeric part of inverse_rle (h263/src/decoder/cpu/rle.rs:
// 112-171), gather (gather.rs:140-204) and the three idct_channel calls (idct.rs:82-201)
// issued at the tail of decode_next_picture (state.rs:432-458).
//
// Workgroup = 256 threads = 4 waves.  Data flow inside a workgroup:
//   load    : the 16 macroblock records of the tile -> LDS; list of "active" blocks
//             (blocks whose class is not trivially Zero) compacted in LDS
//   idct    : rounds of 32 active blocks; 8 lanes per block, lane = one coefficient row.
//             a) coalesced 16-B load of the row, dequant, row pass (T = C x B) -> LDS
//             b) lane = one pixel column: reads its column of T (LDS transposition),
//                column pass, rounding, clamp to [-256,255] -> residual tile in LDS
//   output  : lane = 8 horizontal pixels: half-pel prediction from the reference frame
//             (clamped taps), + residual row from LDS, clip, one 8-byte store; 16
//             consecutive lanes write 128 contiguous bytes of a luma row.
//
// Bit-exactness rules: f32 multiply and add are separately rounded
// (translation unit built with -ffp-contract=off), accumulation order over the
// frequency index is sequential, and the Dc / Vert classes keep their own arithmetic.
#pragma once

#include "dev_common.h"

namespace h263mi {

constexpr int RECON_THREADS = 256;
constexpr int TILE_MBX = 8, TILE_MBY = 2, TILE_MBS = TILE_MBX * TILE_MBY;
constexpr int TILE_TASKS = TILE_MBS * 6;       // 96 blocks
constexpr int ROUND_BLOCKS = RECON_THREADS / 8;  // 32 blocks per idct round
constexpr int TBUF_STRIDE = 72;                // floats per block slot (64 + 8 pad: conflict-free column reads)

struct ReconSmem {
    MbRecord rec[TILE_MBS];
    uint32_t valid_mask;        // bit m: macroblock m of the tile lies inside the picture
    uint32_t act_mask[3];       // bit t: block task t goes through the IDCT
    uint8_t  list[TILE_TASKS];  // compacted active tasks
    float    tbuf[ROUND_BLOCKS * TBUF_STRIDE];
    float    c0buf[ROUND_BLOCKS * 8];
    uint8_t  flags[ROUND_BLOCKS * 8];
    int16_t  res_y[32 * 128];
    int16_t  res_c[2 * 16 * 64];
};

struct TaskId {
    int m, blk;
};

// block task t -> (macroblock in tile, block in macroblock).  Luma tasks are ordered by
// block row then block column so that the 8 blocks of a wave are horizontal neighbours.
H263_HD TaskId task_decode(int t)
{
    TaskId id;
    if (t < 64) {
        int by = t >> 4, bx = t & 15;
        id.m = (by >> 1) * TILE_MBX + (bx >> 1);
        id.blk = ((by & 1) << 1) | (bx & 1);
    } else {
        int c = t - 64;
        id.m = c & 15;
        id.blk = 4 + (c >> 4);
    }
    return id;
}

H263_HD int popc32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(v);
#else
    return __builtin_popcount(v);
#endif
}

// ---- phase 0: records -> LDS -------------------------------------------------------
H263_DEV void recon_phase_load(const ReconArgs &a, ReconSmem &s, int tid, int tile, int pic)
{
    const int tx = tile % (int)a.tiles_x, ty = tile / (int)a.tiles_x;
    if (tid < TILE_MBS * 2) {
        // 32 lanes x 16 B = the tile's 16 records (two 256-B runs, one per macroblock row)
        int m = tid >> 1, half = tid & 1;
        int mbx = tx * TILE_MBX + (m % TILE_MBX), mby = ty * TILE_MBY + (m / TILE_MBX);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (mbx < (int)a.L.mbw && mby < (int)a.L.mbh) {
            const MbRecord *r = a.mbs + (size_t)pic * a.mbs_per_picture + (size_t)mby * a.L.mbw + mbx;
            v = reinterpret_cast<const uint4 *>(r)[half];
        }
        reinterpret_cast<uint4 *>(&s.rec[m])[half] = v;
    }
    if (tid >= 64 && tid < 67) s.act_mask[tid - 64] = 0;
    if (tid == 67) {
        uint32_t vm = 0;
        for (int m = 0; m < TILE_MBS; m++) {
            int mbx = tx * TILE_MBX + (m % TILE_MBX), mby = ty * TILE_MBY + (m / TILE_MBX);
            if (mbx < (int)a.L.mbw && mby < (int)a.L.mbh) vm |= 1u << m;
        }
        s.valid_mask = vm;
    }
}

// ---- phase 1: which blocks need the IDCT -------------------------------------------
H263_DEV void recon_phase_mark(const ReconArgs &, ReconSmem &s, int tid)
{
    if (tid >= TILE_TASKS) return;
    TaskId id = task_decode(tid);
    const MbRecord &r = s.rec[id.m];
    bool valid = (s.valid_mask >> id.m) & 1;
    bool coded = (r.cbp >> id.blk) & 1;
    bool kill = (r.kill >> id.blk) & 1;
    bool intra = mb_is_intra(r.mb_type);
    // coded & kill -> Zero (rle.rs:125-127); uncoded inter -> Zero; uncoded intra -> Dc(level)
    bool active = valid && ((coded && !kill) || (!coded && intra && intradc_level(r.intradc[id.blk]) != 0));
    if (active) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(&s.act_mask[tid >> 5], 1u << (tid & 31));
#else
        s.act_mask[tid >> 5] |= 1u << (tid & 31);
#endif
    }
}

// ---- phase 2: compact the active tasks ----------------------------------------------
H263_DEV void recon_phase_compact(const ReconArgs &, ReconSmem &s, int tid)
{
    if (tid >= TILE_TASKS) return;
    int w = tid >> 5;
    uint32_t mw = s.act_mask[w];
    if (!((mw >> (tid & 31)) & 1)) return;
    int rank = popc32(mw & ((1u << (tid & 31)) - 1u));
    if (w > 0) rank += popc32(s.act_mask[0]);
    if (w > 1) rank += popc32(s.act_mask[1]);
    s.list[rank] = (uint8_t)tid;
}

H263_DEV int recon_n_active(const ReconSmem &s)
{
    return popc32(s.act_mask[0]) + popc32(s.act_mask[1]) + popc32(s.act_mask[2]);
}

// ---- phase 3a: row pass ---------------------------------------------------------------
H263_DEV void recon_phase_idct_rows(const ReconArgs &a, ReconSmem &s, int tid, int pic, int round)
{
    constexpr float B[8][8] = {H263MI_BASIS_ROWS};
    const int slot = tid >> 3, r = tid & 7;
    const int k = round * ROUND_BLOCKS + slot;
    if (k >= recon_n_active(s)) return;
    TaskId id = task_decode(s.list[k]);
    const MbRecord &rec = s.rec[id.m];
    const bool coded = (rec.cbp >> id.blk) & 1;
    const bool intra = mb_is_intra(rec.mb_type);
    const int quant = rec.quant;

    float C[8];
#pragma unroll
    for (int c = 0; c < 8; c++) C[c] = 0.0f;
    if (coded) {
        uint64_t cidx = (a.coeff_base ? a.coeff_base[pic] : 0ull) + rec.coeff_index +
                        (uint64_t)popc32(rec.cbp & ((1u << id.blk) - 1u));
        if (a.coeff_pool_blocks && cidx >= a.coeff_pool_blocks) {
            if (r == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
                atomicOr(a.status, STATUS_COEFF_INDEX_OUT_OF_RANGE);
#else
                *a.status |= STATUS_COEFF_INDEX_OUT_OF_RANGE;
#endif
            }
        } else {
            // 8 lanes x 16 B = one 128-B coefficient block (raster order: lane r holds row r)
            const uint4 raw = *reinterpret_cast<const uint4 *>(a.coeffs + cidx * 64 + (size_t)r * 8);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int c = 0; c < 8; c++) {
                int level = (int)(int16_t)(w[c >> 1] >> ((c & 1) * 16));
                C[c] = (float)dequant_level(level, quant);
            }
        }
    }
    // intra: the DC comes from INTRADC and TCOEFs start at zigzag 1 (rle.rs:117-121)
    if (intra && r == 0) C[0] = (float)intradc_level(rec.intradc[id.blk]);

    // classification inputs (rle.rs:138-149): a non-zero value with y > 0 breaks "horiz",
    // one with x > 0 breaks "vert"
    bool cols_nz = false, row_nz = (C[0] != 0.0f);
#pragma unroll
    for (int c = 1; c < 8; c++) cols_nz = cols_nz || (C[c] != 0.0f);
    row_nz = (row_nz || cols_nz) && (r > 0);
    s.flags[slot * 8 + r] = (uint8_t)((row_nz ? 1 : 0) | (cols_nz ? 2 : 0));
    s.c0buf[slot * 8 + r] = C[0];

    // idct_1d over the coefficient row (idct.rs:52-65): sequential in the frequency index.
    // The leading "0.0 +" is dropped: it can only change the sign of a zero, which never
    // reaches the integer result.
    float T[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float acc = C[0] * B[0][i];
#pragma unroll
        for (int f = 1; f < 8; f++) {
            float p = C[f] * B[f][i];
            acc = acc + p;
        }
        T[i] = acc;
    }
    float4 *dst = reinterpret_cast<float4 *>(&s.tbuf[slot * TBUF_STRIDE + r * 8]);
    dst[0] = make_float4(T[0], T[1], T[2], T[3]);
    dst[1] = make_float4(T[4], T[5], T[6], T[7]);
}

// ---- phase 3b: column pass, rounding, residual tile -------------------------------------
H263_DEV void recon_phase_idct_cols(const ReconArgs &, ReconSmem &s, int tid, int round)
{
    constexpr float B[8][8] = {H263MI_BASIS_ROWS};
    const int slot = tid >> 3, i = tid & 7;
    const int k = round * ROUND_BLOCKS + slot;
    if (k >= recon_n_active(s)) return;
    const int t = s.list[k];

    uint64_t fl;
    memcpy(&fl, &s.flags[slot * 8], 8);
    const bool is_horiz = (fl & 0x0101010101010101ull) == 0;
    const bool is_vert = (fl & 0x0202020202020202ull) == 0;
    const float c00 = s.c0buf[slot * 8];

    // Vert (rle.rs:162-171, idct.rs:152-169) transforms the first column directly; every
    // other class reads column i of the row-pass result (the transposition of idct.rs:171-177).
    const bool vert = is_vert && !is_horiz;
    float col[8];
#pragma unroll
    for (int r = 0; r < 8; r++) col[r] = vert ? s.c0buf[slot * 8 + r] : s.tbuf[slot * TBUF_STRIDE + r * 8 + i];

    float O[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        float acc = col[0] * B[0][j];
#pragma unroll
        for (int r = 1; r < 8; r++) {
            float p = col[r] * B[r][j];
            acc = acc + p;
        }
        O[j] = acc;
    }
    if (vert) {
#pragma unroll
        for (int j = 0; j < 8; j++) O[j] = O[j] * B[0][0];       // idct.rs:160
    }
    const bool dc_class = is_horiz && is_vert;                     // rle.rs:151-160
    if (dc_class) {
#pragma unroll
        for (int j = 0; j < 8; j++) O[j] = c00 * 0.5f;           // idct.rs:119 (exact 0.5, not B00*B00)
    }
    const bool zero_class = dc_class && (c00 == 0.0f);

    int16_t *base;
    int stride;
    if (t < 64) {
        base = &s.res_y[((t >> 4) * 8) * 128 + (t & 15) * 8 + i];
        stride = 128;
    } else {
        int c = t - 64, m = c & 15;
        base = &s.res_c[(c >> 4) * (16 * 64) + ((m >> 3) * 8) * 64 + (m & 7) * 8 + i];
        stride = 64;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        // ((v / 4.0 + signum(v) * 0.5) as i16).clamp(-256, 255)  idct.rs:189-190
        float v = O[j] * 0.25f + (O[j] < 0.0f ? -0.5f : 0.5f);
        v = v < -300.0f ? -300.0f : (v > 300.0f ? 300.0f : v);   // keeps the int conversion in range
        int q = clampi((int)v, -256, 255);
        base[j * stride] = (int16_t)(zero_class ? 0 : q);
    }
}

// ---- motion compensation of one 8-pixel row segment (gather.rs:47-126) -------------------
H263_DEV uint64_t fetch9(const uint8_t *row, int u, int pw, int need, uint32_t &ninth)
{
    uint64_t lo;
    if (u >= 0 && u + need <= pw) {
        lo = load_u64_unaligned(row + u);
        ninth = need > 8 ? row[u + 8] : 0;
    } else {
        lo = 0;
#pragma unroll
        for (int kx = 0; kx < 8; kx++) lo |= (uint64_t)row[clampi(u + kx, 0, pw - 1)] << (8 * kx);
        ninth = row[clampi(u + 8, 0, pw - 1)];
    }
    return lo;
}

H263_DEV uint64_t predict_row(const uint8_t *plane, int pitch, int pw, int ph, int px, int py, int mvx, int mvy)
{
    // HalfPel::into_lerp_parameters (types.rs:721-729): floor(mv / 2), odd -> interpolate
    const int dx = mvx >> 1, ix = mvx & 1, dy = mvy >> 1, iy = mvy & 1;
    const int u = px + dx, v = py + dy;
    const int need = 8 + ix;
    // every tap is clamped to the picture on its own (gather.rs:24-25)
    const uint8_t *r0 = plane + (size_t)clampi(v, 0, ph - 1) * pitch;
    uint32_t n0, n1 = 0;
    uint64_t a = fetch9(r0, u, pw, need, n0);
    uint64_t out = a;
    if (ix | iy) {
        uint64_t a1 = (a >> 8) | ((uint64_t)n0 << 56);
        if (iy) {
            const uint8_t *r1 = plane + (size_t)clampi(v + 1, 0, ph - 1) * pitch;
            uint64_t b = fetch9(r1, u, pw, need, n1);
            if (ix) {
                uint64_t b1 = (b >> 8) | ((uint64_t)n1 << 56);
                out = avg4_u8x8(a, a1, b, b1);          // gather.rs:103-111
            } else {
                out = avg2_u8x8(a, b);                  // gather.rs:115-121
            }
        } else {
            out = avg2_u8x8(a, a1);
        }
    }
    return out;
}

// ---- phase 4: prediction + residual + clip + store ------------------------------------
H263_DEV void recon_phase_output(const ReconArgs &a, ReconSmem &s, int tid, int tile, int pic)
{
    const int tx = tile % (int)a.tiles_x, ty = tile / (int)a.tiles_x;
    const uint8_t *ref = a.ref + (size_t)pic * a.L.frame_bytes;
    uint8_t *cur = a.cur + (size_t)pic * a.L.frame_bytes;

    for (int it = 0; it < 3; it++) {
        const int seg = it * RECON_THREADS + tid;
        int m, blk, task, px, py, pitch, pw, ph;
        size_t plane_off;
        const int16_t *res;
        if (seg < 512) {
            const int yl = seg >> 4, sx = seg & 15;
            m = (yl >> 4) * TILE_MBX + (sx >> 1);
            blk = (((yl >> 3) & 1) << 1) | (sx & 1);
            task = (yl >> 3) * 16 + sx;
            px = tx * (TILE_MBX * 16) + sx * 8;
            py = ty * (TILE_MBY * 16) + yl;
            pitch = (int)a.L.pitch_y; pw = (int)a.L.width; ph = (int)a.L.height;
            plane_off = 0;
            res = &s.res_y[yl * 128 + sx * 8];
        } else {
            const int c = seg - 512, plane = c >> 7, cy = (c & 127) >> 3, csx = c & 7;
            m = (cy >> 3) * TILE_MBX + csx;
            blk = 4 + plane;
            task = 64 + plane * 16 + m;
            px = tx * (TILE_MBX * 8) + csx * 8;
            py = ty * (TILE_MBY * 8) + cy;
            pitch = (int)a.L.pitch_c; pw = (int)a.L.cwidth; ph = (int)a.L.cheight;
            plane_off = plane ? a.L.off_cr : a.L.off_cb;
            res = &s.res_c[plane * (16 * 64) + cy * 64 + csx * 8];
        }
        if (!((s.valid_mask >> m) & 1)) continue;
        const MbRecord &rec = s.rec[m];

        uint64_t pred = 0;                        // intra macroblocks start from zeros (gather.rs:136-138)
        if (mb_is_inter(rec.mb_type)) {
            if (!a.has_ref) {
                // gather.rs:149 Error::UncodedIFrameBlocks -- reported through the status word
                if ((seg & 15) == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
                    atomicOr(a.status, STATUS_INTER_WITHOUT_REFERENCE);
#else
                    *a.status |= STATUS_INTER_WITHOUT_REFERENCE;
#endif
                }
            } else {
                int mvx, mvy;
                if (blk < 4) {
                    mvx = rec.mv[blk][0];
                    mvy = rec.mv[blk][1];
                } else {
                    // gather.rs:182: chroma vector from the i16 sum of the four luma vectors
                    mvx = average_sum_of_mvs(rec.mv[0][0] + rec.mv[1][0] + rec.mv[2][0] + rec.mv[3][0]);
                    mvy = average_sum_of_mvs(rec.mv[0][1] + rec.mv[1][1] + rec.mv[2][1] + rec.mv[3][1]);
                }
                pred = predict_row(ref + plane_off, pitch, pw, ph, px, py, mvx, mvy);
            }
        }

        uint64_t out = pred;
        if ((s.act_mask[task >> 5] >> (task & 31)) & 1) {
            // (clipped_idct + mocomp_pixel).clamp(0, 255)  idct.rs:127-130, 191-194
            uint4 rv = *reinterpret_cast<const uint4 *>(res);
            const uint32_t w[4] = {rv.x, rv.y, rv.z, rv.w};
            out = 0;
#pragma unroll
            for (int kx = 0; kx < 8; kx++) {
                int rr = (int)(int16_t)(w[kx >> 1] >> ((kx & 1) * 16));
                int p = (int)((pred >> (8 * kx)) & 0xff);
                out |= (uint64_t)clampi(p + rr, 0, 255) << (8 * kx);
            }
        }
        *reinterpret_cast<uint64_t *>(cur + plane_off + (size_t)py * pitch + px) = out;
    }
}

}  // namespace h263mi
