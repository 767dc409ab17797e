// msda_tiled.h -- LDS-window MSDeformAttn kernels for gfx950 (encoder-shaped calls, D = 32).
// Placeholder: the tiled kernels are not enabled yet; the direct kernels serve every call.
#pragma once

#include "msda_common.h"

namespace msda {

template <typename T>
bool tiled_fwd_applicable(int, int, int, int, int, int, int, const int64_t *, const int64_t *, const T *, const T *)
{
    return false;
}

template <typename T>
bool tiled_bwd_applicable(int, int, int, int, int, int, int, const int64_t *, const int64_t *, const T *, const T *,
                          const T *)
{
    return false;
}

template <typename T>
hipError_t launch_fwd_tiled(const T *, const int64_t *, const int64_t *, const T *, const T *, T *, int, int, int, int,
                            int, int, int, const int64_t *, const int64_t *, hipStream_t)
{
    return hipErrorNotSupported;
}

template <typename T>
hipError_t launch_bwd_tiled(const T *, const int64_t *, const int64_t *, const T *, const T *, const T *, T *, T *, T *,
                            int, int, int, int, int, int, int, const int64_t *, const int64_t *, hipStream_t)
{
    return hipErrorNotSupported;
}

}  // namespace msda
