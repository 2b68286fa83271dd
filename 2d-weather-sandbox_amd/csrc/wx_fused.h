// wx_fused.h -- fused LDS-tiled kernels (placeholder until the tiled kernels land; see wx_kernels.h)
#pragma once
#include "wx_cells.h"
namespace wx {
struct FusedAIn {
  const float4 *base;
  const char4 *wall;
  const float4 *water, *light, *fb;
  const float2 *dep;
};
struct FusedBIn {
  const float4 *base, *water;
  const char4 *wall;
  const float4 *light;
};
constexpr bool kHaveFused = false;
inline void launch_fused_a(const Geo &, const Uni &, unsigned, const float *, const FusedAIn &, float4 *, float4 *, char4 *, float *, hipStream_t) {}
inline void launch_fused_b(const Geo &, const Uni &, unsigned, const float *, const float *, const float *, const float *, const FusedBIn &, float4 *,
                           float4 *, float4 *, char4 *, float4 *, hipStream_t)
{
}
} // namespace wx
