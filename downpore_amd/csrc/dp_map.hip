// libdownpore_hip.so — map-flavour query (A19 + A20).  Placeholder until the map kernels land.
#include "dp_common.h"

int dp_map_windows_impl(dp_ctx* ctx, const int32_t*, const uint64_t*, uint32_t, int, dp_chain_batch*) {
    return dp_fail(ctx, DP_ERR_STATE, "dp_map_windows: not implemented yet");
}
