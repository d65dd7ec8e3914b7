// port_capi.cpp -- C entry points (vgo_*) of the "port" oracle = oracle/vgo_port.cpp driven by the shared
// batch driver. TEST INFRASTRUCTURE ONLY (see oracle/pyoracle.py for who may load it).
#include "vgo_port.h"
#include <bx/allocator.h>

#define VGO_ENGINE vgo
#define VGO_ENGINE_NAME "port(oracle/vgo_port.cpp restatement, vgmath)"
#define VGO_XFORM vgo::batchTransformPositions
// restatement of vgutil::batchTransformDrawIndices (reference src/vg_util.cpp:447-520, scalar branch :513-518)
static void vgoRebase(const uint16_t* src, uint32_t n, uint16_t* dst, uint16_t delta) { for (uint32_t i = 0; i < n; ++i) { dst[i] = (uint16_t)(src[i] + delta); } }
#define VGO_REBASE(src, n, dst, delta) vgoRebase((src), (n), (dst), (delta))
#include "vgo_driver.inl"
