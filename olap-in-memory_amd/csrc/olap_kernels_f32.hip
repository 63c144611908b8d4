// Kernel instantiations for float cells (one translation unit per cell type keeps the build parallel).
#define OLAP_KERNELS_IMPL
#include "olap_kernels.hpp"

namespace olap {
template struct Launch<float>;
}
