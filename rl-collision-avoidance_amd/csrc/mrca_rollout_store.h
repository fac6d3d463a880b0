// mrca_rollout_store.h -- launchers of the rollout buffer's per-tick stores (mrca_rollout_store.hip); the C-ABI around them
// is in mrca_abi.hip (mrca_rollout_store_state / mrca_rollout_store_outcome, include/mrca_env.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mrca_env.h"
#include "mrca_kernels.h"

namespace mrca {

// one more bit of EnvView::status (mrca_kernels.h holds bits 1, 2, 4): the device-side tick counter of a rollout store was
// outside [0, horizon) -- the launch stored nothing
constexpr uint32_t kStatusBadRolloutRow = 8u;

void launch_rollout_store_state(const EnvView& e, const mrca_rollout_rows& rows, const int64_t* tick, const float* action,
                                const float* logprob, const float* value, hipStream_t s);
void launch_rollout_store_outcome(const EnvView& e, const mrca_rollout_rows& rows, int64_t* tick, uint32_t* ticket, hipStream_t s);

}  // namespace mrca
