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
//   h263::H263StateSet: N of those states advancing together (h263mi_mixed), per-stream behaviour as above
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
    friend class H263StateSet;
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

    // consumer post-processing of the last picture: deblock x3 (strength 0 = off) + BT.601.
    // strength = kStrengthFromHeader: what the picture's own header asks for -- QUANT_TO_STRENGTH[as_header().quantizer] when
    // its USE_DEBLOCKER option is set, else no deblocking (deblock.rs:5-8, picture.rs:61-64, types.rs:94-96, 216)
    static constexpr uint8_t kStrengthFromHeader = H263MI_STRENGTH_FROM_HEADER;
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


// memory of the back-end's GPU (h263mi_device_malloc): where a stream set writes its RGBA pictures
class DeviceBuffer {
public:
    explicit DeviceBuffer(size_t bytes, int device_id = 0) : device_(device_id), bytes_(bytes)
    {
        void *p = nullptr;
        check(h263mi_device_malloc(device_id, bytes, &p));
        p_ = static_cast<uint8_t *>(p);
    }
    ~DeviceBuffer() { (void)h263mi_device_free(device_, p_); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    uint8_t *data() { return p_; }
    size_t size() const { return bytes_; }
    std::vector<uint8_t> download(size_t bytes) const
    {
        if (bytes > bytes_) throw Error(H263MI_ERR_INVALID_ARGUMENT);
        std::vector<uint8_t> out(bytes);
        check(h263mi_device_memcpy_d2h(device_, out.data(), p_, bytes));
        return out;
    }

private:
    int device_ = 0;
    uint8_t *p_ = nullptr;
    size_t bytes_ = 0;
};

// N H263States that advance together on one GPU (h263mi_mixed): every stream is the H263State of state.rs:16-50 -- its own
// last / reference picture, its own picture format (state.rs:157-176: streams of different sizes side by side, a size
// change at an I picture), its own errors -- and one call decodes the next picture of each, one launch per size class.
// There is no such type in the reference (a caller there holds a Vec<H263State> and loops); the per-stream behaviour is the
// reference's, which is what the tests check stream by stream.
class H263StateSet {
public:
    H263StateSet(uint32_t n_streams, uint32_t decoder_options, const h263mi_backend_cfg *cfg = nullptr)
        : n_(n_streams), options_(decoder_options)
    {
        check(h263mi_mixed_create(n_streams, cfg, &m_));
    }
    ~H263StateSet() { h263mi_mixed_destroy(m_); }
    H263StateSet(const H263StateSet &) = delete;
    H263StateSet &operator=(const H263StateSet &) = delete;

    uint32_t len() const { return n_; }
    bool is_sorenson() const { return (options_ & H263MI_SORENSON_SPARK_BITSTREAM) != 0; }

    // what one call did for every stream: `result[s]` is the `Result<(), Error>` of stream s's decode_next_picture
    // (H263MI_OK or the code h263::Error carries), `consumed[s]` the bytes its reader advanced by, `headers[s]` the
    // picture it decoded
    struct Outcome {
        std::vector<int> result;
        std::vector<size_t> consumed;
        std::vector<h263mi_picture_desc> headers;
        bool all_ok() const
        {
            for (int rc : result)
                if (rc != H263MI_OK) return false;
            return true;
        }
    };

    // decode_next_picture (state.rs:138-141) of every stream that has a picture in this call: data[s] == nullptr leaves
    // stream s alone.  d_rgba (optional): per stream a DEVICE buffer of rgba_capacity[s] bytes that receives deblock(strength)
    // (0 = off) + BT.601 of the picture (on a H263MI_CFG_PIPELINE_POST set: by the time of the next call or of sync()).
    // Throws only for a failure of the call itself; a stream's own error is in Outcome::result and leaves that stream as
    // it was (state.rs:142).
    // strength: one for every picture of the call, or H263State::kStrengthFromHeader: every picture with what ITS header
    // asks for (64 streams, 64 quantisers); strengths (optional, ABI 7): the caller's own choice per stream instead.
    Outcome decode_next_pictures(const std::vector<const uint8_t *> &data, const std::vector<size_t> &len, uint8_t strength = 0,
                                 const std::vector<uint8_t *> *d_rgba = nullptr, const std::vector<size_t> *rgba_capacity = nullptr,
                                 uint32_t n_threads = 0, const std::vector<uint8_t> *strengths = nullptr)
    {
        if (data.size() != n_ || len.size() != n_ || (d_rgba && (!rgba_capacity || d_rgba->size() != n_ || rgba_capacity->size() != n_)) ||
            (strengths && strengths->size() != n_))
            throw Error(H263MI_ERR_INVALID_ARGUMENT);
        Outcome o;
        o.result.assign(n_, H263MI_OK);
        o.consumed.assign(n_, 0);
        o.headers.assign(n_, h263mi_picture_desc{});
        check(h263mi_mixed_decode_next_pictures_ps(m_, options_, data.data(), len.data(), o.consumed.data(), n_threads, o.result.data(),
                                                   strength, strengths ? strengths->data() : nullptr, d_rgba ? d_rgba->data() : nullptr,
                                                   rgba_capacity ? rgba_capacity->data() : nullptr, o.headers.data()));
        return o;
    }

    // waits for everything queued (deferred RGBA included); the device's verdict per stream (H263MI_OK, or the error that
    // sent the stream back to its previous picture)
    std::vector<int> sync()
    {
        std::vector<int> rc(n_, H263MI_OK);
        const int first = h263mi_mixed_sync(m_, rc.data());
        bool explained = false;                      // an error return that is some stream's verdict is not thrown
        for (int r : rc) explained = explained || r == first;
        if (first != H263MI_OK && !explained) throw Error(first);
        return rc;
    }

    // get_last_picture (state.rs:61-67) of stream `stream`
    std::optional<DecodedPicture> get_last_picture(uint32_t stream)
    {
        uint16_t w = 0, h = 0;
        const int rc = h263mi_mixed_stream_size(m_, stream, &w, &h);
        if (rc == H263MI_ERR_NO_PICTURE) return std::nullopt;
        check(rc);
        DecodedPicture p;
        p.view_.width = w;
        p.view_.height = h;
        p.view_.chroma_width = (uint16_t)((w + 1) / 2);
        p.view_.chroma_height = (uint16_t)((h + 1) / 2);
        p.luma_.resize((size_t)w * h);
        p.chroma_b_.resize((size_t)p.view_.chroma_width * p.view_.chroma_height);
        p.chroma_r_.resize(p.chroma_b_.size());
        check(h263mi_mixed_copy_yuv(m_, stream, p.luma_.data(), p.chroma_b_.data(), p.chroma_r_.data()));
        return p;
    }

    // H263State::new for one stream (the seeking rule of state.rs:134-137 with the format forgotten too)
    void reset_stream(uint32_t stream) { check(h263mi_mixed_reset_stream(m_, stream)); }
    // how many picture sizes the set has met so far (one fixed-geometry batch and one launch per call each)
    uint32_t size_classes() const { return h263mi_mixed_size_classes(m_); }
    // device memory the frame stores of all sizes together may take (0 = no limit; default: half of the device's memory): a
    // picture of a new size that would go beyond it is that stream's Error (H263MI_ERR_OUT_OF_MEMORY)
    void set_memory_limit(uint64_t bytes) { check(h263mi_mixed_set_memory_limit(m_, bytes)); }
    uint64_t frame_store_bytes() const { return h263mi_mixed_frame_store_bytes(m_); }
    h263mi_mixed *raw() { return m_; }

private:
    h263mi_mixed *m_ = nullptr;
    uint32_t n_ = 0, options_ = 0;
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
