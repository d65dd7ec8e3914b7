// vgx_cmdlist.hip -- the reference's command-list byte-code as input. Host code only.
//
// A vg::CommandList is a HOST byte buffer (CommandList::m_CommandBuffer, reference src/vg.cpp:243-247, 5694-5723):
// records of {CommandHeader{uint32 type, uint32 size}, padded to 16 bytes}{payload, padded to 16 bytes}, written by the cl*
// functions (:2403-2690) and replayed by ctxSubmitCommandList's switch (:4332-4625) through the same ctxXXX calls an
// immediate-mode caller makes. vgx_cmdlist_decode walks that buffer ONCE and produces what the batch entry points take:
//   - one path per ctxBeginPath .. group (vgx_pathset_desc arrays: the cl* payloads are the pathXXX arguments verbatim), and
//   - one vgx_draw per FillPathColor / StrokePathColor with the state the interpreter would have at that point folded in:
//     transform and m_AvgScale (PushState / PopState / Transform* / SetViewBox, :3934-4122, updateState :4927-4944), the
//     colour's alpha scaled by the global alpha, the stroke width scaled / clamped and the Thin switch (:3401-3433).
// The walk is sequential by nature (variable-size records, a state stack) and the buffer lives in host memory; decoding it
// on the host next to vgx_pathset_create's validation costs nothing measurable (a memory-speed pass, once per recorded
// list, not per frame) -- uploading it to decode with dependent 16-byte loads would be slower and still need the host pass.
// What vgx_tessellate cannot express is counted in num_skipped and otherwise ignored: gradient / image fills and strokes,
// IndexedTriList, clip and scissor commands, text, nested command lists, concave fills (libtess2 stays with the caller:
// vgx_concave_*).
// Parity unpinned: vg.cpp needs bgfx and cannot be compiled here; the byte layout and the state arithmetic are restated
// from the cited lines and pinned by hand-assembled streams in tests/test_cmdlist.py.
#include <hip/hip_runtime.h>
#include "../../include/vgx.h"
#include "vgmath.h"
#include <string.h>
#include <vector>

namespace {

// CommandType::Enum, vg.cpp:177-241
enum {
	CT_BeginPath = 0, CT_MoveTo, CT_LineTo, CT_CubicTo, CT_QuadraticTo, CT_ArcTo, CT_Arc, CT_Rect, CT_RoundedRect, CT_RoundedRectVarying,
	CT_Circle, CT_Ellipse, CT_Polyline, CT_ClosePath,
	CT_FillPathColor, CT_FillPathGradient, CT_FillPathImagePattern, CT_StrokePathColor, CT_StrokePathGradient, CT_StrokePathImagePattern,
	CT_IndexedTriList,
	CT_BeginClip, CT_EndClip, CT_ResetClip, CT_CreateLinearGradient, CT_CreateBoxGradient, CT_CreateRadialGradient, CT_CreateImagePattern,
	CT_PushState, CT_PopState, CT_ResetScissor, CT_SetScissor, CT_IntersectScissor,
	CT_TransformIdentity, CT_TransformScale, CT_TransformTranslate, CT_TransformRotate, CT_TransformMult, CT_SetViewBox, CT_SetGlobalAlpha,
	CT_Text, CT_TextBox, CT_SubmitCommandList, CT_Count_
};
const uint32_t kAlign = 16;       // VG_CONFIG_COMMAND_LIST_ALIGNMENT, vg.cpp:40
const uint32_t kHeaderSize = 16;  // alignSize(sizeof(CommandHeader), 16), vg.cpp:708

struct St { float m[6]; float avgScale; float alpha; };

void updateState(St& s) // vg.cpp:4927-4935
{
	const float sx = vgm_sqrt(s.m[0] * s.m[0] + s.m[2] * s.m[2]);
	const float sy = vgm_sqrt(s.m[1] * s.m[1] + s.m[3] * s.m[3]);
	s.avgScale = (sx + sy) * 0.5f;
}

void mul3(const float* a, const float* b, float* r) // vgutil::multiplyMatrix3, vg_util.h:36-44
{
	r[0] = a[0] * b[0] + a[2] * b[1];
	r[1] = a[1] * b[0] + a[3] * b[1];
	r[2] = a[0] * b[2] + a[2] * b[3];
	r[3] = a[1] * b[2] + a[3] * b[3];
	r[4] = a[0] * b[4] + a[2] * b[5] + a[4];
	r[5] = a[1] * b[4] + a[3] * b[5] + a[5];
}

float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Builder
{
	vgx_cmdlist_out* out;
	bool store;
	uint32_t npaths, ncmd, nargs, ndraws, nskipped;
	// current path = commands [pathCmd0, ncmd); referenced = a draw already points at it
	uint32_t pathCmd0, pathArg0;
	bool havePath, referenced, overflow;
	void pathCmd(uint8_t type, const float* a, uint32_t n)
	{
		if (store) {
			if (ncmd >= out->cap_cmds || nargs + n > out->cap_args) { overflow = true; return; }
			out->cmd_type[ncmd] = type;
			memcpy(out->args + nargs, a, n * sizeof(float));
			out->cmd_arg_off[ncmd + 1] = nargs + n;
		}
		++ncmd; nargs += n;
	}
	void beginPath()
	{
		closePathRecord();
		havePath = true; referenced = false;
		pathCmd0 = ncmd; pathArg0 = nargs;
	}
	void closePathRecord() // the current path is complete: path_cmd_begin[npaths + 1]
	{
		if (!havePath) { return; }
		if (store) {
			if (npaths >= out->cap_paths) { overflow = true; } else { out->path_cmd_begin[npaths + 1] = ncmd; }
		}
		++npaths;
		havePath = false;
	}
};

} // namespace

extern "C" int vgx_cmdlist_decode(const void* bytes, uint32_t size, const vgx_cmdlist_state* st0, vgx_cmdlist_out* out)
{
	if ((!bytes && size) || !st0 || !out || (size % kAlign) != 0) {
		return VGX_E_INVALID_ARG;
	}
	const bool store = out->cmd_type && out->cmd_arg_off && out->args && out->path_cmd_begin && out->draws; // else: count only
	Builder B;
	memset(&B, 0, sizeof(B));
	B.out = out; B.store = store;
	if (store) { out->cmd_arg_off[0] = 0; out->path_cmd_begin[0] = 0; }

	std::vector<St> stack(1);
	memcpy(stack[0].m, st0->mtx, sizeof(float) * 6);
	stack[0].alpha = st0->global_alpha;
	updateState(stack[0]);
	// per-command argument counts of the current path (count pass of a fork needs them without stored arrays)
	std::vector<uint32_t> curArgCounts;
	float pathScale = stack[0].avgScale; // pathReset / strokerReset scale of the current path (ctxBeginPath, vg.cpp:2969-2981)

	const uint8_t* p = (const uint8_t*)bytes;
	const uint8_t* end = p + size;
	while (p < end) {
		if ((uint32_t)(end - p) < kHeaderSize) { return VGX_E_INVALID_ARG; }
		uint32_t type, psize;
		memcpy(&type, p, 4); memcpy(&psize, p + 4, 4);
		p += kHeaderSize;
		if ((psize % kAlign) != 0 || psize > (uint32_t)(end - p) || type >= CT_Count_) { return VGX_E_INVALID_ARG; }
		const uint8_t* d = p;
		p += psize;
		const float* f = (const float*)d;
		St& S = stack.back();
		auto need = [&](uint32_t n) { return psize >= n; };
		auto pathArgs = [&](uint8_t vt, uint32_t nfloats) -> int {
			if (!need(nfloats * 4)) { return VGX_E_INVALID_ARG; }
			if (!B.havePath) { ++B.nskipped; return VGX_OK; } // path command before any BeginPath: the reference would append to a stale path
			// a path command after a fill / stroke of the same path (no BeginPath in between): the reference keeps appending
			// to the Path object, so the next fill / stroke sees all of it. Paths are immutable here: continue on a copy.
			if (B.referenced) {
				const uint32_t c0 = B.pathCmd0, c1 = B.ncmd;
				B.closePathRecord();
				B.havePath = true; B.referenced = false;
				B.pathCmd0 = B.ncmd; B.pathArg0 = B.nargs;
				for (uint32_t c = c0; c < c1; ++c) {
					const uint32_t n = curArgCounts[c - c0];
					if (store && !B.overflow) {
						std::vector<float> tmp(out->args + out->cmd_arg_off[c], out->args + out->cmd_arg_off[c] + n);
						B.pathCmd(out->cmd_type[c], tmp.data(), n);
					} else { ++B.ncmd; B.nargs += n; }
				}
			}
			float tmp[8];
			memcpy(tmp, f, nfloats * 4);
			B.pathCmd(vt, tmp, nfloats);
			curArgCounts.push_back(nfloats);
			return VGX_OK;
		};
		int rc = VGX_OK;
		switch (type) {
		case CT_BeginPath: B.beginPath(); curArgCounts.clear(); pathScale = S.avgScale; break;
		case CT_MoveTo: rc = pathArgs(VGX_CMD_MOVE_TO, 2); break;
		case CT_LineTo: rc = pathArgs(VGX_CMD_LINE_TO, 2); break;
		case CT_CubicTo: rc = pathArgs(VGX_CMD_CUBIC_TO, 6); break;
		case CT_QuadraticTo: rc = pathArgs(VGX_CMD_QUAD_TO, 4); break;
		case CT_ArcTo: rc = pathArgs(VGX_CMD_ARC_TO, 5); break;
		case CT_Arc: { // five floats + Winding::Enum (vg.cpp:2459-2472); vgx: sixth argument 1 = CW
			if (!need(24)) { return VGX_E_INVALID_ARG; }
			uint32_t dir; memcpy(&dir, d + 20, 4);
			float a[6]; memcpy(a, f, 20); a[5] = dir == 1u ? 1.0f : 0.0f;
			const float* keep = f; f = a; rc = pathArgs(VGX_CMD_ARC, 6); f = keep;
		} break;
		case CT_Rect: rc = pathArgs(VGX_CMD_RECT, 4); break;
		case CT_RoundedRect: rc = pathArgs(VGX_CMD_ROUNDED_RECT, 5); break;
		case CT_RoundedRectVarying: rc = pathArgs(VGX_CMD_ROUNDED_RECT_VARYING, 8); break;
		case CT_Circle: rc = pathArgs(VGX_CMD_CIRCLE, 3); break;
		case CT_Ellipse: rc = pathArgs(VGX_CMD_ELLIPSE, 4); break;
		case CT_ClosePath: rc = pathArgs(VGX_CMD_CLOSE, 0); break;
		case CT_Polyline: { // uint32 numPoints + coordinates (vg.cpp:2551-2559)
			if (!need(4)) { return VGX_E_INVALID_ARG; }
			uint32_t np; memcpy(&np, d, 4);
			if (np == 0 || np > (psize - 4) / 8) { return VGX_E_INVALID_ARG; }
			if (!B.havePath) { ++B.nskipped; break; }
			if (B.referenced) { const float* keep = f; rc = pathArgs(VGX_CMD_CLOSE, 0); f = keep; if (rc == VGX_OK) { --B.ncmd; curArgCounts.pop_back(); } } // fork only
			std::vector<float> pts((size_t)np * 2);
			memcpy(pts.data(), d + 4, (size_t)np * 8);
			B.pathCmd(VGX_CMD_POLYLINE, pts.data(), np * 2);
			curArgCounts.push_back(np * 2);
		} break;
		case CT_FillPathColor: { // uint32 flags, Color (vg.cpp:2619-2627); ctxFillPathColor :3061-3179
			if (!need(8)) { return VGX_E_INVALID_ARG; }
			uint32_t flags, color; memcpy(&flags, d, 4); memcpy(&color, d + 4, 4);
			const uint32_t a = (uint32_t)(uint8_t)(S.alpha * (float)(color >> 24));
			if (a == 0 || !B.havePath) { if (a != 0) { ++B.nskipped; } break; } // transparent: the reference returns before any geometry
			if (flags & 0x01u) { ++B.nskipped; break; } // PathType::Concave: libtess2 (vgx_concave_*)
			if (store) {
				if (B.ndraws >= out->cap_draws) { B.overflow = true; }
				else {
					vgx_draw& dr = out->draws[B.ndraws];
					memset(&dr, 0, sizeof(dr));
					dr.path = B.npaths; // index of the current (not yet closed) path
					dr.fill_flags = VGX_FILL_ENABLE | ((flags & 0x04u) ? VGX_FILL_AA : 0u);
					dr.fill_color = (color & 0x00FFFFFFu) | (a << 24);
					dr.scale = pathScale; dr.tess_tol = st0->tess_tol; dr.fringe = st0->fringe;
					memcpy(dr.mtx, S.m, sizeof(float) * 6);
				}
			}
			++B.ndraws; B.referenced = true;
		} break;
		case CT_StrokePathColor: { // float width, uint32 flags, Color (vg.cpp:2660-2669); ctxStrokePathColor :3401-3433
			if (!need(12)) { return VGX_E_INVALID_ARG; }
			float width; uint32_t flags, color; memcpy(&width, d, 4); memcpy(&flags, d + 4, 4); memcpy(&color, d + 8, 4);
			const float fringe = st0->fringe;
			const float scaled = (flags & (1u << 5)) ? width : clampf(width * S.avgScale, 0.0f, 200.0f); // StrokeFlags::FixedWidth
			const bool thin = scaled <= fringe;
			const float c = clampf(scaled, 0.0f, fringe);
			const float alphaScale = !thin ? S.alpha : S.alpha * (c * c);
			const uint32_t a = (uint32_t)(uint8_t)(alphaScale * (float)(color >> 24));
			if (a == 0 || !B.havePath) { if (a != 0) { ++B.nskipped; } break; }
			if (store) {
				if (B.ndraws >= out->cap_draws) { B.overflow = true; }
				else {
					vgx_draw& dr = out->draws[B.ndraws];
					memset(&dr, 0, sizeof(dr));
					dr.path = B.npaths;
					const bool aa = (flags & 0x10u) != 0;
					dr.stroke_flags = VGX_STROKE_FLAGS((flags >> 2) & 3u, flags & 3u, aa, thin && aa);
					dr.stroke_color = (color & 0x00FFFFFFu) | (a << 24);
					dr.stroke_width = thin ? fringe : scaled;
					dr.scale = pathScale; dr.tess_tol = st0->tess_tol; dr.fringe = fringe;
					memcpy(dr.mtx, S.m, sizeof(float) * 6);
				}
			}
			++B.ndraws; B.referenced = true;
		} break;
		case CT_PushState: { const St top = stack.back(); stack.push_back(top); } break; // vg.cpp:3934-3943
		case CT_PopState: if (stack.size() > 1) { stack.pop_back(); } else { return VGX_E_INVALID_ARG; } break;
		case CT_TransformIdentity: S.m[0] = 1; S.m[1] = 0; S.m[2] = 0; S.m[3] = 1; S.m[4] = 0; S.m[5] = 0; updateState(S); break;
		case CT_TransformScale: if (!need(8)) { return VGX_E_INVALID_ARG; } S.m[0] = f[0] * S.m[0]; S.m[1] = f[0] * S.m[1]; S.m[2] = f[1] * S.m[2]; S.m[3] = f[1] * S.m[3]; updateState(S); break; // :4044-4053
		case CT_TransformTranslate: if (!need(8)) { return VGX_E_INVALID_ARG; } S.m[4] += S.m[0] * f[0] + S.m[2] * f[1]; S.m[5] += S.m[1] * f[0] + S.m[3] * f[1]; updateState(S); break; // :4055-4062
		case CT_TransformRotate: { // :4064-4082, bx::cos / bx::sin = the pinned ones of vgmath.h
			if (!need(4)) { return VGX_E_INVALID_ARG; }
			const float c = vgm_cos(f[0]), s = vgm_sin(f[0]);
			float m[6];
			m[0] = c * S.m[0] + s * S.m[2]; m[1] = c * S.m[1] + s * S.m[3];
			m[2] = -s * S.m[0] + c * S.m[2]; m[3] = -s * S.m[1] + c * S.m[3];
			m[4] = S.m[4]; m[5] = S.m[5];
			memcpy(S.m, m, sizeof(m)); updateState(S);
		} break;
		case CT_TransformMult: { // six floats + TransformOrder::Enum (Pre = 0, Post = 1), :4084-4100
			if (!need(28)) { return VGX_E_INVALID_ARG; }
			uint32_t order; memcpy(&order, d + 24, 4);
			float m[6], r[6]; memcpy(m, f, 24);
			if (order == 1u) { mul3(S.m, m, r); } else { mul3(m, S.m, r); }
			memcpy(S.m, r, sizeof(r)); updateState(S);
		} break;
		case CT_SetViewBox: { // :4102-4121 needs the canvas size
			if (!need(16)) { return VGX_E_INVALID_ARG; }
			const float sxv = st0->canvas_width / f[2], syv = st0->canvas_height / f[3];
			S.m[0] = sxv * S.m[0]; S.m[1] = sxv * S.m[1]; S.m[2] = syv * S.m[2]; S.m[3] = syv * S.m[3];
			S.m[4] -= S.m[0] * f[0] + S.m[2] * f[1]; S.m[5] -= S.m[1] * f[0] + S.m[3] * f[1];
			updateState(S);
		} break;
		case CT_SetGlobalAlpha: if (!need(4)) { return VGX_E_INVALID_ARG; } S.alpha = f[0]; break;
		default: ++B.nskipped; break; // gradients, images, IndexedTriList, clip, scissor, text, nested lists
		}
		if (rc != VGX_OK) { return rc; }
	}
	B.closePathRecord();
	out->num_paths = B.npaths; out->num_cmds = B.ncmd; out->num_args = B.nargs; out->num_draws = B.ndraws; out->num_skipped = B.nskipped;
	if (store && B.overflow) { return VGX_E_NOSPACE; }
	return VGX_OK;
}
