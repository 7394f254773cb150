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

// Fork / join of up to SIDE_LANES side streams per device for INDEPENDENT kernels of one call (sage_conv_backward: the
// bias-gradient column sums and the grad_x chain beside the weight-gradient GEMM -- each a few launches of 5-15 us that
// would otherwise queue up behind one another).  fork(): every lane waits for what `main` has enqueued so far; join():
// `main` waits for every lane.  The same event pattern as SideCopy: it survives stream capture into a HIP graph, where
// the lanes become parallel branches.
constexpr int SIDE_LANES = 2;
struct LaneSet;

class SideLanes {
  public:
    int fork(hipStream_t main, int lanes);
    hipStream_t lane(int i) const;
    int join(hipStream_t main);
    bool forked() const { return set_ != nullptr; }

  private:
    LaneSet *set_ = nullptr;
    int lanes_ = 0;
    std::unique_lock<std::mutex> hold_;
};

int enqueue_copy_features(const float *x, int32_t F, float *out, int64_t out_cols, int64_t N, hipStream_t stream);   // SideCopy's kernel on any stream

extern int g_copy_batches_per_wave;            // pope_debug_set(POPE_KNOB_COPY_BATCHES)

}  // namespace pope
