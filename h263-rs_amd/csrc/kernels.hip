// kernels.hip -- __global__ wrappers and launchers of the gfx950 kernels.
// Built with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math).
#include "kernels.h"

#include "post_kernel.inl"
#include "recon_kernel.inl"
#include "synth.inl"

namespace h263mi {

// ---------------------------------------------------------------------------------------
// k_recon: grid = (tiles per picture, pictures), 256 threads
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(RECON_THREADS) void k_recon(ReconArgs a)
{
    __shared__ __attribute__((aligned(16))) ReconSmem s;
    const int tid = threadIdx.x, tile = blockIdx.x, pic = blockIdx.y;
    recon_phase_load(a, s, tid, tile, pic);
    __syncthreads();
    recon_phase_mark(a, s, tid);
    __syncthreads();
    recon_phase_compact(a, s, tid);
    __syncthreads();
    const int n_active = recon_n_active(s);
    for (int round = 0; round * ROUND_BLOCKS < n_active; round++) {
        recon_phase_idct_rows(a, s, tid, pic, round);
        __syncthreads();
        recon_phase_idct_cols(a, s, tid, round);
        __syncthreads();
    }
    recon_phase_output(a, s, tid, tile, pic);
}

hipError_t launch_recon(const ReconArgs &args, hipStream_t stream)
{
    dim3 grid(args.tiles_x * args.tiles_y, args.n_pictures, 1);
    hipLaunchKernelGGL(k_recon, grid, dim3(RECON_THREADS), 0, stream, args);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// k_post: grid = (tiles per picture, pictures), 256 threads
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(POST_THREADS) void k_post(PostArgs a)
{
    __shared__ __attribute__((aligned(16))) PostSmem s;
    const int tid = threadIdx.x, tile = blockIdx.x, pic = blockIdx.y;
    post_phase_load(a, s, tid, tile, pic);
    __syncthreads();
    if (a.strength) {
        post_phase_hedges(a, s, tid, tile);
        __syncthreads();
        post_phase_vedges(a, s, tid, tile);
        __syncthreads();
    }
    post_phase_store(a, s, tid, tile, pic);
}

hipError_t launch_post(const PostArgs &args, hipStream_t stream)
{
    dim3 grid(args.tiles_x * args.tiles_y, args.n_pictures, 1);
    hipLaunchKernelGGL(k_post, grid, dim3(POST_THREADS), 0, stream, args);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// synthetic record generators (bench / test support)
// ---------------------------------------------------------------------------------------
__global__ void k_synth_headers(SynthArgs a)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_streams * a.mbs_per_picture) return;
    const uint32_t p = g / a.mbs_per_picture, mb = g % a.mbs_per_picture;
    MbRecord r = synth_mb_header(a.kind, a.first_stream_id + p, a.frame_idx, mb);
    a.mbs[g] = r;
    a.counts[g] = (uint32_t)__popc(r.cbp);
}

// exclusive scan of the per-macroblock coded-block counts of one picture -> coeff_index
__global__ __launch_bounds__(1024) void k_synth_scan(SynthArgs a)
{
    __shared__ uint32_t sums[1024];
    const uint32_t p = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = a.mbs_per_picture, chunk = (n + 1023) / 1024;
    const uint32_t lo = tid * chunk, hi = lo + chunk < n ? lo + chunk : n;
    uint32_t local = 0;
    for (uint32_t i = lo; i < hi; i++) local += a.counts[p * n + i];
    sums[tid] = local;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = tid >= off ? sums[tid - off] : 0;
        __syncthreads();
        sums[tid] += v;
        __syncthreads();
    }
    uint32_t run = sums[tid] - local;     // exclusive prefix of this thread's chunk
    for (uint32_t i = lo; i < hi; i++) {
        a.mbs[p * n + i].coeff_index = run;
        run += a.counts[p * n + i];
    }
    if (tid == 1023) a.totals[p] = sums[1023];
}

__global__ void k_synth_coeffs(SynthArgs a)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_streams * a.mbs_per_picture * 6) return;
    const uint32_t blk = g % 6, gm = g / 6, p = gm / a.mbs_per_picture, mb = gm % a.mbs_per_picture;
    const MbRecord r = a.mbs[gm];
    if (!((r.cbp >> blk) & 1)) return;
    int16_t c[64];
    synth_block_coeffs(a.kind, a.first_stream_id + p, a.frame_idx, mb, (int)blk, c);
    const uint64_t idx = a.coeff_base[p] + r.coeff_index + (uint64_t)__popc(r.cbp & ((1u << blk) - 1u));
    uint4 *dst = reinterpret_cast<uint4 *>(a.coeffs + idx * 64);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        uint32_t w[4];
        for (int k = 0; k < 4; k++) w[k] = (uint16_t)c[q * 8 + 2 * k] | ((uint32_t)(uint16_t)c[q * 8 + 2 * k + 1] << 16);
        dst[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

hipError_t launch_synth_headers(const SynthArgs &a, hipStream_t stream)
{
    const uint32_t n = a.n_streams * a.mbs_per_picture;
    hipLaunchKernelGGL(k_synth_headers, dim3((n + 255) / 256), dim3(256), 0, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_synth_scan, dim3(a.n_streams), dim3(1024), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_synth_coeffs(const SynthArgs &a, hipStream_t stream)
{
    const uint32_t n = a.n_streams * a.mbs_per_picture * 6;
    hipLaunchKernelGGL(k_synth_coeffs, dim3((n + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

}  // namespace h263mi
