// ref_capi.cpp -- C wrapper that drives the REFERENCE'S OWN compiled sources (oracle/_ref/libvgref.so).
// TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this. It is compiled together with /root/reference/src/{path,stroker,vg_util}.cpp (+libtess2,
// so stroker.cpp links) by oracle/Makefile; no reference source is copied into this repository.
// kind = "reference" in bench.py's cpu_baseline.
#include <vg/path.h>
#include <vg/stroker.h>
#include "vg_util.h" // /root/reference/src/vg_util.h
#include <bx/allocator.h>

#define VGO_ENGINE vg
#define VGO_ENGINE_NAME "reference(vg-renderer src @ /root/reference, scalar build, bx_shim+vgmath)"
#define VGO_XFORM vgutil::batchTransformPositions
#define VGO_REBASE(src, n, dst, delta) vgutil::batchTransformDrawIndices((src), (n), (dst), (delta)) // the reference's own (vg_util.cpp:447-520)
#define VGO_INVERT3(t, inv) vgutil::invertMatrix3((t), (inv))                      // vg_util.cpp:14-33
#include "vgo_driver.inl"
