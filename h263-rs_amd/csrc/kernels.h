// kernels.h -- host-visible launchers of the gfx950 kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "dev_common.h"

namespace h263mi {

struct SynthArgs {
    int kind;
    uint32_t n_streams, first_stream_id, frame_idx, mbs_per_picture;
    uint32_t stream_stride;     // picture p is stream first_stream_id + p * stream_stride (1: consecutive ids)
    MbRecord *mbs;              // device, n_streams * mbs_per_picture
    uint32_t *counts;           // device scratch, one per macroblock
    uint32_t *totals;           // device, coded blocks per picture
    int16_t *coeffs;            // device pool
    const uint64_t *coeff_base; // device, per picture
};

// geometry of the merged launch (k_frame), made by its launcher
struct FrameGeom {
    uint32_t groups;            // groups of 32 luma rows per picture
    uint32_t recon_per_group;   // reconstruction waves per group: 2 per 8x2-macroblock tile (one per macroblock row)
    uint32_t post_per_group;    // post tiles (waves) per group
    uint32_t inv_per_group;     // ceil(2^32 / (recon_per_group + post_per_group))
    uint32_t flip;              // walk the pictures of the batch in descending order
    uint32_t bands;             // XCDs that share one picture (8, 4 or 2): 8 / bands pictures side by side
};
// `words` (HOST pointer, n_pictures STREAM_* words, or nullptr): the streams of the launch differ -- each picture's waves read
// its word.  Up to STREAM_WORDS_INLINE pictures they travel in the kernel arguments; beyond that the caller has put them into
// device memory (args.stream_state) and passes nullptr here.
hipError_t launch_recon(const ReconArgs &args, hipStream_t stream, const uint32_t *words = nullptr);
// k_recon over `rargs` and k_post over `pargs` (same number of pictures, same picture size) as ONE launch.
// `descending`: walk the pictures last to first.  A caller that alternates the direction from one frame index to the
// next reads the planes it wrote last -- the ones still in the infinity cache -- first (2 % on a 64-stream batch).
hipError_t launch_frame(const ReconArgs &rargs, const PostArgs &pargs, hipStream_t stream, bool descending, const uint32_t *words = nullptr);
hipError_t launch_post(const PostArgs &args, hipStream_t stream, const uint32_t *words = nullptr);
hipError_t launch_synth_headers(const SynthArgs &args, hipStream_t stream);
hipError_t launch_synth_coeffs(const SynthArgs &args, hipStream_t stream);
// streaming probes: mode 0 copy in -> out, 1 read in (out = 16-byte sink), 2 write out; bytes is a multiple of 16;
// shape < probe_shapes(mode) picks the launch shape (kernels.hip)
int probe_shapes(int mode);
const char *probe_shape_name(int mode, int shape);
hipError_t launch_probe(int mode, int shape, const void *in, void *out, size_t bytes, hipStream_t stream);

}  // namespace h263mi
