// vgx_pathset_host.h -- host-side validation of the command grammar (see include/vgx.h) and derivation of the static
// per-command structure the kernels use (sub-path heads / tails, serial-path flag). Plain C++. Round 6: vgx_pathset_create
// builds these tables on the DEVICE (vgx_pathset.hip); what is left of this file in the product is the validator
// (vgx_pathset_validate, and the slow path that names the status of an invalid set). The table builders below are the ORACLE of
// the device build: csrc/vgx_hosttest.cpp exports them (vgxt_pathset_table), tests/test_gpu_pathset_build.py compares every
// table of every set byte for byte.
#ifndef VGX_PATHSET_HOST_H
#define VGX_PATHSET_HOST_H

#include "vgx_internal_types.h"
#include <vector>
#include <math.h>

static const int kArgCount[VGX_CMD_COUNT_] = { 2, 2, 6, 4, 0, 5, 6, 4, 5, 8, 3, 4, -1 };

static int vgx_pathset_validate_host(const vgx_pathset_desc* d, std::vector<uint8_t>* cmdFlags, std::vector<uint32_t>* spStart, std::vector<uint8_t>* pathFlags, uint32_t* maxCmds)
{
	if (!d || !d->path_cmd_begin || (d->ncmd && (!d->cmd_type || !d->cmd_arg_off)) || !d->cmd_arg_off) {
		return VGX_E_INVALID_ARG;
	}
	if (d->path_cmd_begin[0] != 0 || d->path_cmd_begin[d->npaths] != d->ncmd || d->cmd_arg_off[0] != 0) {
		return VGX_E_INVALID_ARG;
	}
	if (d->ncmd >= 0x7FFFFFFFu) { // bit 31 of a command index carries a flag in the draw window
		return VGX_E_INVALID_ARG;
	}
	cmdFlags->assign(d->ncmd, 0);
	spStart->assign(d->ncmd, 0);
	pathFlags->assign(d->npaths ? d->npaths : 1, 0);
	*maxCmds = 0;
	const uint32_t nargs = d->cmd_arg_off[d->ncmd];
	if (nargs && !d->args) {
		return VGX_E_INVALID_ARG;
	}
	for (uint32_t i = 0; i < nargs; ++i) {
		if (!isfinite(d->args[i])) {
			return VGX_E_NONFINITE;
		}
	}
	for (uint32_t p = 0; p < d->npaths; ++p) {
		const uint32_t c0 = d->path_cmd_begin[p], c1 = d->path_cmd_begin[p + 1];
		if (c1 < c0 || c1 > d->ncmd) {
			return VGX_E_INVALID_ARG;
		}
		if (c1 - c0 > *maxCmds) { *maxCmds = c1 - c0; }
		bool open = false; // a sub-path is open and may take more vertices
		uint32_t head = c0;
		for (uint32_t c = c0; c < c1; ++c) {
			const uint32_t t = d->cmd_type[c];
			if (t >= VGX_CMD_COUNT_) {
				return VGX_E_INVALID_ARG;
			}
			if (d->cmd_arg_off[c + 1] < d->cmd_arg_off[c]) {
				return VGX_E_INVALID_ARG;
			}
			const uint32_t na = d->cmd_arg_off[c + 1] - d->cmd_arg_off[c];
			if (t == VGX_CMD_POLYLINE) {
				if (na < 2 || (na & 1)) { return VGX_E_INVALID_ARG; }
			} else if ((int)na != kArgCount[t]) {
				return VGX_E_INVALID_ARG;
			}
			const bool isShape = t >= VGX_CMD_RECT && t <= VGX_CMD_ELLIPSE;
			bool starts = false;
			if (t == VGX_CMD_MOVE_TO || isShape) {
				starts = true;
			} else if (t == VGX_CMD_ARC) {
				// pathArc wraps its angles with `while (a > 2pi) a -= 2pi` loops (path.cpp:637-652): beyond ~1e8 the
				// subtraction no longer changes a float and the reference spins forever; keep them where the loops
				// are short (the same loops run on the device, bit for bit)
				const float* aa = d->args + d->cmd_arg_off[c];
				if (fabsf(aa[3]) > 1.0e5f || fabsf(aa[4]) > 1.0e5f) { return VGX_E_INVALID_ARG; }
				starts = !open; // pathArc: moveTo when there is no open sub-path, else lineTo (path.cpp:663-667)
				if (!open && c != c0) {
					// a leading arc is only well defined at the very start of a path or after MOVE_TO-less state;
					// after CLOSE / a closed shape the reference would append to a closed sub-path
					return VGX_E_INVALID_PATH;
				}
			} else if (!open) {
				return VGX_E_INVALID_PATH; // LINE_TO/CUBIC_TO/... need an open sub-path (path.cpp:82,88)
			}
			if (starts) { head = c; }
			(*spStart)[c] = head;
			if (starts) { (*cmdFlags)[c] |= VGX_CF_STARTS_SUB; }
			if (t == VGX_CMD_ARC || t == VGX_CMD_ARC_TO || isShape) { (*pathFlags)[p] |= VGX_PF_SERIAL; }
			open = !(t == VGX_CMD_CLOSE || isShape);
		}
		for (uint32_t c = c0; c < c1; ++c) {
			const bool last = (c + 1 == c1);
			if (last) { (*cmdFlags)[c] |= VGX_CF_LAST_IN_PATH | VGX_CF_LAST_IN_SUB; }
			else {
				if ((*cmdFlags)[c + 1] & VGX_CF_STARTS_SUB) { (*cmdFlags)[c] |= VGX_CF_LAST_IN_SUB; }
				if (d->cmd_type[c + 1] == VGX_CMD_CLOSE) { (*cmdFlags)[c] |= VGX_CF_NEXT_IS_CLOSE; }
			}
		}
	}
	return VGX_OK;
}

// The 64-byte command records (VgxCmdRec): what vgx_pathset_create's kernel k_ps_rec writes, as the host loop it replaced.
static inline void vgx_pathset_records_host(const vgx_pathset_desc* desc, const uint8_t* cmdFlags, const uint32_t* spStart, VgxCmdRec* rec)
{
	for (uint32_t c = 0; c < desc->ncmd; ++c) {
		VgxCmdRec& r = rec[c];
		const uint32_t ao = desc->cmd_arg_off[c];
		r.type = desc->cmd_type[c];
		r.flags = cmdFlags[c];
		r.na = desc->cmd_arg_off[c + 1] - ao;
		r.arg_off = ao;
		r.start[0] = ao >= 2 ? desc->args[ao - 2] : 0.0f;
		r.start[1] = ao >= 2 ? desc->args[ao - 1] : 0.0f;
		for (uint32_t i = 0; i < 8; ++i) { r.a[i] = (i < r.na && r.type != VGX_CMD_POLYLINE) ? desc->args[ao + i] : 0.0f; }
		if (r.type <= VGX_CMD_CLOSE || r.type == VGX_CMD_POLYLINE) {
			// first point of the command's sub-path (its MOVE_TO): pathClose's last-vs-first test (path.cpp:716-722)
			// is evaluated by the CLOSE lane and by the lane in front of it
			const uint32_t hc = spStart[c];
			if (desc->cmd_type[hc] == VGX_CMD_MOVE_TO) {
				const uint32_t ho = desc->cmd_arg_off[hc];
				r.a[6] = desc->args[ho]; r.a[7] = desc->args[ho + 1];
			}
		}
		r.pad[0] = 0.0f; r.pad[1] = 0.0f;
	}
}

// Per path: the commands that end a sub-path (k_flatten_gather walks these instead of every command)
static inline void vgx_pathset_subs_host(const vgx_pathset_desc* desc, const uint8_t* cmdFlags, std::vector<uint32_t>* pathSubBegin, std::vector<uint32_t>* subLastCmd)
{
	pathSubBegin->assign(desc->npaths + 1, 0);
	subLastCmd->clear();
	for (uint32_t p = 0; p < desc->npaths; ++p) {
		(*pathSubBegin)[p] = (uint32_t)subLastCmd->size();
		for (uint32_t c = desc->path_cmd_begin[p]; c < desc->path_cmd_begin[p + 1]; ++c) {
			if (cmdFlags[c] & VGX_CF_LAST_IN_SUB) { subLastCmd->push_back(c - desc->path_cmd_begin[p]); }
		}
	}
	(*pathSubBegin)[desc->npaths] = (uint32_t)subLastCmd->size();
}

#endif
