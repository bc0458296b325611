// Extra translation unit of libvft_hip.so: explicit instances of the line-search kernels (k_ml_quartet, double) - by far the largest
// device code of the library - so that the build is several parallel hipcc jobs.  vft_api.hip declares the same
// instances `extern` and launches them; the kernels never call across units.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "vft_layout.h"
#include "vft_device.h"
#include "vft_kernels_profile.h"
#include "vft_kernels_ml.h"

VFT_ML_QUARTET_INSTANCES_F64()

#ifdef VFT_ML_TIMING
extern "C" int vft_ml_ticks(unsigned long long *out) {
    return (int) hipMemcpyFromSymbol(out, HIP_SYMBOL(vftMlTicks), 16 * sizeof(unsigned long long));
}
#endif
