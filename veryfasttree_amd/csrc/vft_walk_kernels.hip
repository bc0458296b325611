// Extra translation unit of libvft_hip.so: explicit instances of the walk server (vft_kernels_walk.h), so that the build stays
// several parallel hipcc jobs.  vft_api.hip declares the same instances `extern` and launches them.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "vft_layout.h"
#include "vft_device.h"
#include "vft_kernels_profile.h"
#include "vft_kernels_walk.h"

VFT_WALK_SERVER_INSTANCES()
