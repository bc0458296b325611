// Extra translation unit of libvft_hip.so: explicit instances of the line-search kernels for alignments beyond the register-resident
// instances' 2 048 columns (vft_kernels_ml_long.h).  vft_api.hip declares the same instances `extern` and launches them.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "vft_layout.h"
#include "vft_device.h"
#include "vft_kernels_profile.h"
#include "vft_kernels_ml_long.h"

VFT_ML_LONG_INSTANCES()
