// vgx_pathset_dev.h -- vgx_pathset_create on the device (round 6): the caller's raw SoA arrays go up as they are, every derived
// table is built by kernels (vgx_pathset.hip). What the reference does per command while a path is recorded (pathMoveTo /
// pathLineTo / pathClose bookkeeping, path.cpp:62-84, 684-726, 761-784) and what vgx_pathset_host.h / vgx_thin.h restate as host
// loops (kept as the ORACLE of these kernels: csrc/vgx_hosttest.cpp, tests/test_gpu_pathset_build.py compares every table byte
// for byte).
#ifndef VGX_PATHSET_DEV_H
#define VGX_PATHSET_DEV_H

#include "vgx_internal.h"
#include "vgx_thin.h"
#include "vgx_scan.h"

struct VgxPsTotals // zeroed before the build; read by the host once, at the end
{
	uint32_t err;            // != 0: some check failed (the host validator then names the status, vgx_pathset_validate)
	uint32_t maxCmds;        // longest path, in commands
	uint32_t hasEmpty;       // a path without commands
	uint32_t hasSerial;      // a path with ARC / ARC_TO / closed shapes (exact serial builder)
	uint32_t notThin;        // a command other than MOVE_TO / LINE_TO / CLOSE
	uint32_t nsubs;          // sub-paths of the set (LAST_IN_SUB commands)
	uint32_t thinIneligible; // a path with more than 65 536 sub-paths (vgx_thin.h keeps the ordinal in 16 bits)
	uint32_t pad;
};

struct VgxPsM // monoid of the scan over commands: (max, max, +, or)
{
	uint32_t head1; // 1 + index of the last command that starts a sub-path (0: none yet)
	uint32_t path1; // 1 + index of the last path that begins at or before the command (0: none yet)
	uint32_t nlast; // LAST_IN_SUB commands
	uint32_t f;     // own item: flags | type << 8 | serial << 16 | notThin << 17 (the scan's value of this word is not used)
};

struct VgxPsBuild
{
	// raw arrays, already in their final places inside the blob
	const uint8_t* type; const uint32_t* argOff; const float* args; const uint32_t* pcb;
	uint32_t ncmd, npaths, nargs;
	// derived tables (blob)
	uint32_t* spStart; uint8_t* flags; uint8_t* pathFlags; VgxCmdRec* rec; VgxCmdThin* thin /* record of command 0 */;
	uint32_t* subBegin; uint32_t* subLast; VgxThinPath* tp; VgxThinSub* ts;
	// temporaries (context scratch)
	uint32_t* pathAt;    // [ncmd + 1] zeroed by the host together with `tot` (they are neighbours); p + 1 at the first command of every non-empty path
	uint32_t* pathOf;    // [ncmd]
	uint32_t* lastSubEx; // [ncmd + 1] LAST_IN_SUB commands in front of the command
	uint32_t* nvEx;      // [ncmd + 1] thin sets: polyline vertices in front of the command (pops included)
	VgxPsM* partialA;    // [VGX_MSCAN_BLOCKS]
	Sum3* partialB;      // [VGX_SCAN_BLOCKS]
	VgxPsTotals* tot;
};

// Frame-sized sets go up as ONE image of their raw arrays, their blobs are recycled through the context, and the host looks at their
// opcodes (a few KB) to leave the thin passes out when some command is a curve.
static inline bool vgx_pathset_is_small(uint32_t ncmd, uint32_t npaths, uint32_t nargs) { return ncmd <= 16384u && npaths <= 16384u && nargs <= 131072u; }

void vgx_launch_pathset_build(const VgxPsBuild& a, bool maybeThin, hipStream_t s);

#endif
