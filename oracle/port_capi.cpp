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
// restatement of vgutil::invertMatrix3 (reference src/vg_util.cpp:14-33)
static void vgoInvert3(const float* t, float* inv)
{
	const double det = (double)t[0] * t[3] - (double)t[2] * t[1];
	if (det > -1e-6 && det < 1e-6) {
		inv[0] = inv[2] = 1.0f;
		inv[1] = inv[3] = inv[4] = inv[5] = 0.0f;
		return;
	}
	const double invdet = 1.0 / det;
	inv[0] = (float)(t[3] * invdet);
	inv[2] = (float)(-t[2] * invdet);
	inv[4] = (float)(((double)t[2] * t[5] - (double)t[3] * t[4]) * invdet);
	inv[1] = (float)(-t[1] * invdet);
	inv[3] = (float)(t[0] * invdet);
	inv[5] = (float)(((double)t[1] * t[4] - (double)t[0] * t[5]) * invdet);
}
#define VGO_INVERT3(t, inv) vgoInvert3((t), (inv))
#include "vgo_driver.inl"
