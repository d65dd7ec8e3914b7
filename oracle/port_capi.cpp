// port_capi.cpp -- C entry points (vgo_*) of the "port" oracle = oracle/vgo_port.cpp driven by the shared
// batch driver. TEST INFRASTRUCTURE ONLY (see oracle/pyoracle.py for who may load it).
#include "vgo_port.h"
#include <bx/allocator.h>

#define VGO_ENGINE vgo
#define VGO_ENGINE_NAME "port(oracle/vgo_port.cpp restatement, vgmath)"
#define VGO_XFORM vgo::batchTransformPositions
#include "vgo_driver.inl"
