// out[:, :F] = x (utils.py:177 concat_into_features) as a kernel of its own on a per-device side stream: forked from the
// caller's stream, running BESIDE the MFMA tile kernel of the node2vec embedding instead of inside or after it, joined
// at the end of the call.  The copy is pure HBM streaming (357 MB at configs[2]); the tile kernel is bound by the matrix
// cores.  (Beside the BFS levels of the geodesic path it does not pay: they are bound by the fabric the copy streams
// through -- pope_geodesic_run in geodesic.hip.)
#pragma once

#include <mutex>

#include "common.h"

namespace pope {

struct SideStream;

class SideCopy {
  public:
    // 16-byte pieces, 32-bit byte offsets, at most 16 pieces per lane and row
    static bool eligible(const float *x, int32_t F, const float *out, int64_t out_cols, int64_t N);
    int fork(hipStream_t main);                                                        // marks the point on `main` the copy has to wait for (everything enqueued so far)
    int launch(const float *x, int32_t F, float *out, int64_t out_cols, int64_t N);    // the copy kernel on the side stream, behind the fork point
    int join(hipStream_t main);                                                        // `main` waits for the copy
    bool forked() const { return side_ != nullptr; }

  private:
    SideStream *side_ = nullptr;
    std::unique_lock<std::mutex> hold_;        // held from fork() until this object dies: the event pair is shared by the device's callers
};

int enqueue_copy_features(const float *x, int32_t F, float *out, int64_t out_cols, int64_t N, hipStream_t stream);   // SideCopy's kernel on any stream

extern int g_copy_batches_per_wave;            // pope_debug_set(POPE_KNOB_COPY_BATCHES)

}  // namespace pope
