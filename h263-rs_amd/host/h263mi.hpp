// h263mi.hpp -- C++ mirror of the reference's public API over the C ABI (include/h263mi.h).
//
// The reference is a Rust library; Rust is not available in this image, so the host side above
// the C ABI is C++.  Names, argument meaning and error behaviour follow the reference:
//
//   h263::H263State::{new_, is_sorenson, get_last_picture, get_reference_picture,
//                     cleanup_buffers, decode_next_picture}      h263/src/decoder/state.rs:40-490
//   h263::DecodedPicture::{as_luma, as_chroma_b, as_chroma_r, luma_samples_per_row,
//                          chroma_samples_per_row, as_yuv}       h263/src/decoder/picture.rs:61-142
//   h263::DecoderOption                                          h263/src/decoder/types.rs:3-17
//   h263::Error                                                  h263/src/error.rs:6-93
//   deblock::deblock, deblock::QUANT_TO_STRENGTH                 deblock/src/deblock.rs:5-8,305-315
//   yuv::bt601::yuv420_to_rgba                                   yuv/src/bt601.rs:105-196
//
// Rust `Result<T, Error>` becomes a thrown h263::Error; `Option<&DecodedPicture>` becomes
// std::optional<DecodedPicture> (a host copy of the planes, valid independently of the state).
#pragma once

#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/h263mi.h"

namespace h263 {

struct Error : std::runtime_error {
    int code;
    explicit Error(int c) : std::runtime_error(h263mi_strerror(c)), code(c) {}
    // error.rs:66-93
    bool is_eof_error() const { return code == H263MI_ERR_UNHANDLED_IO_ERROR; }
    bool is_macroblock_error() const
    {
        return code == H263MI_ERR_INVALID_MACROBLOCK_HEADER || code == H263MI_ERR_INVALID_MACROBLOCK_CODED_BITS;
    }
    bool is_gob_error() const { return code == H263MI_ERR_INVALID_GOB_HEADER; }
};

inline void check(int rc)
{
    if (rc != H263MI_OK) throw Error(rc);
}

// decoder/types.rs:3-17
namespace DecoderOption {
constexpr uint32_t SORENSON_SPARK_BITSTREAM = H263MI_SORENSON_SPARK_BITSTREAM;
constexpr uint32_t USE_SCALABILITY_MODE = H263MI_USE_SCALABILITY_MODE;
}  // namespace DecoderOption

// picture.rs:8-58: exact-size, tightly packed planes
class DecodedPicture {
public:
    const std::vector<uint8_t> &as_luma() const { return luma_; }
    const std::vector<uint8_t> &as_chroma_b() const { return chroma_b_; }
    const std::vector<uint8_t> &as_chroma_r() const { return chroma_r_; }
    size_t luma_samples_per_row() const { return view_.width; }
    size_t chroma_samples_per_row() const { return view_.chroma_width; }
    std::tuple<const std::vector<uint8_t> &, const std::vector<uint8_t> &, const std::vector<uint8_t> &> as_yuv() const
    {
        return {luma_, chroma_b_, chroma_r_};
    }
    const h263mi_frame_view &as_header() const { return view_; }

private:
    friend class H263State;
    h263mi_frame_view view_{};
    std::vector<uint8_t> luma_, chroma_b_, chroma_r_;
};

class H263State {
public:
    // H263State::new(decoder_options)  state.rs:42-50
    explicit H263State(uint32_t decoder_options, const h263mi_backend_cfg *cfg = nullptr)
    {
        check(h263mi_state_new(decoder_options, cfg, &s_));
    }
    ~H263State() { h263mi_state_free(s_); }
    H263State(const H263State &) = delete;
    H263State &operator=(const H263State &) = delete;

    bool is_sorenson() const { return h263mi_state_is_sorenson(s_) != 0; }     // state.rs:53-56
    void cleanup_buffers() { check(h263mi_state_cleanup_buffers(s_)); }        // state.rs:81-98
    void reset() { check(h263mi_state_reset(s_)); }                            // seeking rule, state.rs:134-137

    // state.rs:61-67
    std::optional<DecodedPicture> get_last_picture() const { return fetch(&h263mi_get_last_picture); }
    // state.rs:72-78 (returns the last picture whenever a reference exists, like the reference)
    std::optional<DecodedPicture> get_reference_picture() const { return fetch(&h263mi_get_reference_picture); }

    // state.rs:138-141 over one coded picture held in memory; returns the bytes consumed
    size_t decode_next_picture(const uint8_t *data, size_t len)
    {
        size_t used = 0;
        check(h263mi_decode_next_picture(s_, data, len, &used));
        return used;
    }

    // H263State::parse_picture (state.rs:102-111): header peek
    h263mi_picture_desc parse_picture(const uint8_t *data, size_t len) const
    {
        h263mi_picture_desc d{};
        check(h263mi_parse_picture_header(s_, data, len, &d));
        return d;
    }

    // record-level form of decode_next_picture (state.rs:421-483): what the host parser hands over
    void submit_picture(const h263mi_picture_desc &desc, const std::vector<h263mi_mb_record> &mbs,
                        const std::vector<int16_t> &coeffs)
    {
        check(h263mi_submit_picture(s_, &desc, mbs.data(), mbs.size(), coeffs.data(), coeffs.size() / 64));
    }

    // the same with sparse coefficient transport: one event (level << 16 | x + 8y) per non-zero LEVEL
    void submit_picture_events(const h263mi_picture_desc &desc, const std::vector<h263mi_mb_record> &mbs,
                               const std::vector<uint32_t> &block_first_event, const std::vector<uint32_t> &events)
    {
        check(h263mi_submit_picture_events(s_, &desc, mbs.data(), mbs.size(), block_first_event.data(),
                                           block_first_event.empty() ? 0 : block_first_event.size() - 1, events.data(),
                                           events.size()));
    }

    // consumer post-processing of the last picture: deblock x3 (strength 0 = off) + BT.601
    std::vector<uint8_t> render_rgba(uint8_t strength) const
    {
        h263mi_frame_view v;
        check(h263mi_get_last_picture(s_, &v));
        std::vector<uint8_t> rgba((size_t)v.width * v.height * 4);
        check(h263mi_render_rgba(s_, strength, rgba.data()));
        return rgba;
    }

    // the same straight into page-locked memory of the caller (h263mi_host_alloc / h263mi_host_register): the buffer a
    // renderer reuses for every picture instead of the fresh Vec<u8> of bt601.rs:128; `rgba` holds width * height * 4 bytes
    void render_rgba_into_pinned(uint8_t strength, uint8_t *rgba) const { check(h263mi_render_rgba_pinned(s_, strength, rgba)); }

    h263mi_state *raw() { return s_; }

private:
    std::optional<DecodedPicture> fetch(int (*getter)(const h263mi_state *, h263mi_frame_view *)) const
    {
        DecodedPicture p;
        int rc = getter(s_, &p.view_);
        if (rc == H263MI_ERR_NO_PICTURE) return std::nullopt;
        check(rc);
        p.luma_.resize((size_t)p.view_.width * p.view_.height);
        p.chroma_b_.resize((size_t)p.view_.chroma_width * p.view_.chroma_height);
        p.chroma_r_.resize(p.chroma_b_.size());
        check(h263mi_copy_yuv(s_, p.luma_.data(), p.chroma_b_.data(), p.chroma_r_.data()));
        return p;
    }
    h263mi_state *s_ = nullptr;
};

// page-locked, device-visible host memory (h263mi_host_alloc) for H263State::render_rgba_into_pinned
class PinnedBuffer {
public:
    explicit PinnedBuffer(size_t bytes) : bytes_(bytes)
    {
        void *p = nullptr;
        check(h263mi_host_alloc(bytes, &p));
        p_ = static_cast<uint8_t *>(p);
    }
    ~PinnedBuffer() { (void)h263mi_host_free(p_); }
    PinnedBuffer(const PinnedBuffer &) = delete;
    PinnedBuffer &operator=(const PinnedBuffer &) = delete;
    uint8_t *data() { return p_; }
    size_t size() const { return bytes_; }

private:
    uint8_t *p_ = nullptr;
    size_t bytes_ = 0;
};

}  // namespace h263

namespace deblock {
// pub const QUANT_TO_STRENGTH: [u8; 32]  deblock.rs:5-8
inline const uint8_t (&QUANT_TO_STRENGTH)[32] = h263mi_quant_to_strength;

// pub fn deblock(data: &[u8], width: usize, strength: u8) -> Vec<u8>  deblock.rs:305-315
inline std::vector<uint8_t> deblock(const std::vector<uint8_t> &data, size_t width, uint8_t strength)
{
    std::vector<uint8_t> out(data.size());
    h263::check(h263mi_deblock(data.data(), data.size(), width, strength, out.data()));
    return out;
}
}  // namespace deblock

namespace yuv {
namespace bt601 {
// pub fn yuv420_to_rgba(y, chroma_b, chroma_r, y_width) -> Vec<u8>  bt601.rs:105-196
inline std::vector<uint8_t> yuv420_to_rgba(const std::vector<uint8_t> &y, const std::vector<uint8_t> &chroma_b,
                                           const std::vector<uint8_t> &chroma_r, size_t y_width)
{
    if (chroma_b.size() != chroma_r.size()) throw h263::Error(H263MI_ERR_INVALID_ARGUMENT);
    std::vector<uint8_t> rgba(y.size() * 4);
    h263::check(h263mi_bt601_yuv420_to_rgba(y.data(), y.size(), chroma_b.data(), chroma_r.data(), chroma_b.size(),
                                            y_width, rgba.data()));
    return rgba;
}
}  // namespace bt601
}  // namespace yuv
