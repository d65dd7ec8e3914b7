// vgx_cmdlist.hip -- the reference's command-list byte-code as input. Host code only.
//
// A vg::CommandList is a HOST byte buffer (CommandList::m_CommandBuffer, reference src/vg.cpp:243-247, 5694-5723):
// records of {CommandHeader{uint32 type, uint32 size}, padded to 16 bytes}{payload, padded to 16 bytes}, written by the cl*
// functions (:2403-2967) and replayed by ctxSubmitCommandList's switch (:4273-4637) through the same ctxXXX calls an
// immediate-mode caller makes. vgx_cmdlist_decode walks that buffer ONCE and produces what the batch entry points take:
//   - one path per ctxBeginPath .. group (vgx_pathset_desc arrays: the cl* payloads are the pathXXX arguments verbatim), and
//   - one vgx_draw per fill / stroke command (colour, gradient and image-pattern paint alike: they call the same strokerXXX
//     functions and differ in the createDrawCommand_XXX they end in, :3061-3668) with the state the interpreter would have at
//     that point folded in: transform and m_AvgScale (PushState / PopState / Transform* / SetViewBox, :3934-4122, updateState
//     :4927-4944; the transform a path is drawn with is the one of its FIRST fill / stroke, transformPath :4957-4975), the
//     colour's alpha scaled by the global alpha, the stroke width scaled / clamped and the Thin switch, the scissor and clip
//     state the draw command would carry, and vgx_draw::state_key = what allocDrawCommand compares before merging.
// The walk is sequential by nature (variable-size records, a state stack) and the buffer lives in host memory; decoding it
// on the host next to vgx_pathset_create's validation costs nothing measurable (a memory-speed pass, once per recorded
// list, not per frame) -- uploading it to decode with dependent 16-byte loads would be slower and still need the host pass.
// Parity: pinned against the reference's own writers and interpreter (the test suite compiles src/vg.cpp unmodified behind
// a recording bgfx stand-in): tests/test_cmdlist_ref.py records lists with vg::clXxx, decodes the bytes here and compares the
// frame with what vg::submitCommandList + vg::end produce.
#include <hip/hip_runtime.h>
#include "../../include/vgx.h"
#include "vgmath.h"
#include <string.h>
#include <vector>

static_assert(sizeof(vgx_draw_state) == 24 && sizeof(vgx_paint) == 96 && sizeof(vgx_cmdlist_ref) == 16, "vgx.h layout");

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
enum { DT_Textured = 0, DT_ColorGradient = 1, DT_ImagePattern = 2, DT_Clip = 3 }; // DrawCommand::Type, vg.cpp:110-118
const uint32_t kAlign = 16;       // VG_CONFIG_COMMAND_LIST_ALIGNMENT, vg.cpp:40
const uint32_t kHeaderSize = 16;  // alignSize(sizeof(CommandHeader), 16), vg.cpp:708
const uint32_t kBlack = 0xFF000000u; // Colors::Black

struct St { float m[6]; float scissor[4]; float alpha; float avgScale; }; // vg::State, vg.cpp:62-69

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

void invert3(const float* t, float* inv) // vgutil::invertMatrix3, vg_util.cpp:14-33 (double precision inside)
{
	const double det = (double)t[0] * t[3] - (double)t[2] * t[1];
	if (det > -1e-6 && det < 1e-6) {
		inv[0] = 1.0f; inv[1] = 0.0f; inv[2] = 1.0f; inv[3] = 0.0f; inv[4] = 0.0f; inv[5] = 0.0f; // sic: the reference's "identity"
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

inline float clampf(float v, float lo, float hi) { return vgm_max(vgm_min(v, hi), lo); } // bx::clamp = max(min(a, hi), lo)
inline uint32_t setAlpha(uint32_t c, uint8_t a) { return (c & 0x00FFFFFFu) | ((uint32_t)a << 24); } // colorSetAlpha, vg.inl:95-98

struct Decoder
{
	const vgx_cmdlist_state* st0;
	vgx_cmdlist_out* out;
	bool store;
	bool overflow;
	uint32_t npaths, ncmd, nargs, ndraws, nskipped, npaints;
	// the Path object: commands [pathCmd0, ncmd) once BeginPath was seen
	bool havePath;       // a BeginPath group is open
	bool transformed;    // Context::m_PathTransformed: transformPath ran for this path (latches pathMtx)
	float pathMtx[6];
	float pathScale;     // pathReset / strokerReset scale (ctxBeginPath, vg.cpp:2969-2981)
	std::vector<St> stack;
	// draw-command bookkeeping
	bool forceNew;       // m_ForceNewDrawCommand / m_ForceNewClipCommand raised since the last draw
	uint32_t generation;
	bool haveLastScissor;
	uint16_t lastScissor[4]; // scissor of the last draw COMMAND (= of the last non-clip draw emitted)
	vgx_paint dummyPaint;    // where newPaint() writes when nothing is stored
	// clip state (ClipState, vg.cpp:71-76) in units of draws
	bool recordClip;
	uint32_t clipRule, clipFirst, clipNum;
	uint32_t drawBase; // this decode's draw i is draw drawBase + i of the frame (vgx_cmdlist_state::draw_base)
	uint32_t nextGradient, nextImagePattern, maxGradients, maxImagePatterns;
	uint32_t depth, maxDepth;

	St& S() { return stack.back(); }

	void pathCmd(uint8_t type, const float* a, uint32_t n)
	{
		if (store) {
			if (ncmd >= out->cap_cmds || nargs + n > out->cap_args) { overflow = true; }
			else {
				out->cmd_type[ncmd] = type;
				if (n) { memcpy(out->args + nargs, a, n * sizeof(float)); }
				out->cmd_arg_off[ncmd + 1] = nargs + n;
			}
		}
		++ncmd; nargs += n;
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
	void beginPath()
	{
		closePathRecord();
		havePath = true; transformed = false;
		pathScale = S().avgScale;
		subPts = 0; pathMaxPts = 0; pathCurved = false;
	}
	// transformPath (vg.cpp:4957-4975): the first caller fixes the matrix the path's vertices are transformed with
	void latch() { if (!transformed) { memcpy(pathMtx, S().m, sizeof(pathMtx)); transformed = true; } }

	uint32_t subPts = 0, pathMaxPts = 0; bool pathCurved = false; // vertices of the current path's line-only sub-paths (current one / largest finished one)
	float subFirst[2] = { 0, 0 }, subLast[2] = { 0, 0 };
	uint32_t rawColor = 0; // Color operand of the paint command being decoded (vgx_draw_state::raw_color)
	bool emitHasMesh = false; // the draw being emitted certainly allocates a draw command (IndexedTriList)
	uint32_t ntriM = 0, ntriV = 0, ntriI = 0; // IndexedTriList meshes / vertices / indices so far
	void emit(uint32_t type, uint32_t handle, uint32_t fillFlags, uint32_t fillColor, uint32_t strokeFlags, uint32_t strokeColor, float strokeWidth)
	{
		const St& s = S();
		uint16_t sc[4];
		for (int i = 0; i < 4; ++i) { sc[i] = (uint16_t)s.scissor[i]; }
		if (forceNew) { ++generation; forceNew = false; }
		if (store) {
			if (ndraws >= out->cap_draws) { overflow = true; }
			else {
				vgx_draw& dr = out->draws[ndraws];
				memset(&dr, 0, sizeof(dr));
				dr.path = npaths; // index of the current (not yet closed) path
				dr.fill_flags = fillFlags; dr.fill_color = fillColor;
				dr.stroke_flags = strokeFlags; dr.stroke_color = strokeColor; dr.stroke_width = strokeWidth;
				dr.scale = pathScale; dr.tess_tol = st0->tess_tol; dr.fringe = st0->fringe;
				memcpy(dr.mtx, pathMtx, sizeof(float) * 6);
				dr.state_key = (generation << 20) | (type << 16) | (handle & 0xFFFFu);
				if (out->draw_state) {
					vgx_draw_state& ds = out->draw_state[ndraws];
					memcpy(ds.scissor, sc, sizeof(sc));
					if (type == DT_Clip) { ds.clip_rule = 0; ds.clip_first_draw = 0xFFFFFFFFu; ds.clip_num_draws = 0; } // allocClipCommand, vg.cpp:5449-5450
					// a non-clip draw INSIDE an open region (gradient / image-pattern paints do not look at m_RecordClipCommands) sees the
					// region still empty: ctxBeginClip sets m_NumCmds = 0, ctxEndClip fills it in (vg.cpp:3670-3697)
					else { ds.clip_rule = clipRule; ds.clip_first_draw = clipFirst; ds.clip_num_draws = recordClip ? 0u : clipNum; }
					ds.raw_color = rawColor;
				}
			}
		}
		// PopState compares the restored scissor with the frame's last draw COMMAND (vg.cpp:3950-3965), and a draw whose path
		// yields no mesh allocates none (a filled 2-point path ...). The meshes are not known here; what is known are the vertex
		// counts of sub-paths made of line segments only (pathArgs) -- draws that certainly have no mesh do not count.
		// (A curve that flattens to fewer vertices than the stroker's minimum is not seen here.)
		const bool isFill = fillFlags != 0;
		const uint32_t mostPts = subPts > pathMaxPts ? subPts : pathMaxPts;
		const bool noMesh = !emitHasMesh && !pathCurved && mostPts < (isFill ? 3u : 2u);
		if (type != DT_Clip && !noMesh) { memcpy(lastScissor, sc, sizeof(sc)); haveLastScissor = true; }
		++ndraws;
	}

	vgx_paint* newPaint(uint32_t type, uint32_t handle, const float* inv)
	{
		vgx_paint* p = nullptr;
		vgx_paint& dummy = dummyPaint; // count pass / overflow: a member, not a function-local static (two threads may decode at once)
		if (store && out->paints) {
			if (npaints >= out->cap_paints) { overflow = true; p = &dummy; } else { p = &out->paints[npaints]; }
		} else { p = &dummy; }
		++npaints;
		memset(p, 0, sizeof(*p));
		p->type = type; p->handle = handle;
		p->matrix[0] = inv[0]; p->matrix[1] = inv[1]; p->matrix[2] = 0.0f;
		p->matrix[3] = inv[2]; p->matrix[4] = inv[3]; p->matrix[5] = 0.0f;
		p->matrix[6] = inv[4]; p->matrix[7] = inv[5]; p->matrix[8] = 1.0f;
		return p;
	}
	static void colors(vgx_paint* p, uint32_t icol, uint32_t ocol) // colorGetRed.. / 255.0f, vg.cpp:3764-3771
	{
		for (int i = 0; i < 4; ++i) {
			p->inner_color[i] = (float)((icol >> (8 * i)) & 0xFFu) / 255.0f;
			p->outer_color[i] = (float)((ocol >> (8 * i)) & 0xFFu) / 255.0f;
		}
	}

	int run(const uint8_t* p, uint32_t size, uint32_t listFlags);
};

int Decoder::run(const uint8_t* p, uint32_t size, uint32_t listFlags)
{
	// ctxSubmitCommandList, vg.cpp:4273-4330
	if (depth >= maxDepth) { return VGX_OK; }
	++depth;
	const bool hasCache = (listFlags & VGX_CL_CACHEABLE) != 0;                    // getCommandListCacheStackTop() != nullptr
	const bool cullCmds = !hasCache && (listFlags & VGX_CL_ALLOW_CULLING) != 0;   // :4299-4300
	const uint32_t firstGradientID = nextGradient & 0xFFFFu, firstImagePatternID = nextImagePattern & 0xFFFFu;
	bool skipCmds = false;
	const float fringe = st0->fringe;
	const uint8_t* end = p + size;
	while (p < end) {
		if ((uint32_t)(end - p) < kHeaderSize) { return VGX_E_INVALID_ARG; }
		uint32_t type, psize;
		memcpy(&type, p, 4); memcpy(&psize, p + 4, 4);
		p += kHeaderSize;
		if ((psize % kAlign) != 0 || psize > (uint32_t)(end - p) || type >= CT_Count_) { return VGX_E_INVALID_ARG; }
		const uint8_t* d = p;
		p += psize;
		if (skipCmds && type >= CT_FillPathColor && type <= CT_StrokePathImagePattern) { continue; } // :4335-4338
		const float* f = (const float*)d;
		auto need = [&](uint32_t n) { return psize >= n; };
		auto u32at = [&](uint32_t off) { uint32_t v; memcpy(&v, d + off, 4); return v; };
		auto u16at = [&](uint32_t off) { uint16_t v; memcpy(&v, d + off, 2); return v; };
		auto f32at = [&](uint32_t off) { float v; memcpy(&v, d + off, 4); return v; };
		// a path command: legal while the path has not been transformed yet (VG_CHECK(!m_PathTransformed), :2984-3059)
		auto pathArgs = [&](uint8_t vt, const float* a, uint32_t nfloats) {
			if (!havePath || transformed) { ++nskipped; return; }
			pathCmd(vt, a, nfloats);
			// vertices of the sub-paths made of line segments only, counted the way vg::Path does (pathAddVertex's epsilon test
			// path.cpp:769-775, pathPolyline's on its first point only :684-704, pathClose's pop :707-726), see emit()
			auto nearPt = [](float ax, float ay, float bx, float by) { const float dx = ax - bx, dy = ay - by; return dx * dx + dy * dy < VGM_EPSILON; };
			if (vt == VGX_CMD_MOVE_TO) {
				if (subPts > pathMaxPts) { pathMaxPts = subPts; } // the previous sub-path is finished
				subPts = 1; subFirst[0] = subLast[0] = a[0]; subFirst[1] = subLast[1] = a[1];
			} else if (vt == VGX_CMD_LINE_TO) {
				if (subPts == 0 || !nearPt(subLast[0], subLast[1], a[0], a[1])) { ++subPts; subLast[0] = a[0]; subLast[1] = a[1]; }
			} else if (vt == VGX_CMD_POLYLINE && nfloats >= 2) {
				uint32_t np = nfloats / 2;
				if (subPts > 0 && nearPt(subLast[0], subLast[1], a[0], a[1])) { --np; }
				if (np > 0) { subPts += np; subLast[0] = a[nfloats - 2]; subLast[1] = a[nfloats - 1]; }
			} else if (vt == VGX_CMD_CLOSE) {
				if (subPts > 2 && nearPt(subLast[0], subLast[1], subFirst[0], subFirst[1])) { --subPts; }
			} else { pathCurved = true; } // curves, arcs, shapes: any number of vertices
		};
		auto globalAlpha = [&]() { return hasCache ? 1.0f : S().alpha; };
		// alpha of the colour a Color / ImagePattern fill or stroke hands to the stroker (:3071-3075 and siblings)
		auto scaledColor = [&](uint32_t color, float alphaScale) { return setAlpha(color, (uint8_t)(alphaScale * (uint8_t)(color >> 24))); };
		// stroke width rules shared by the three strokePath flavours (:3416-3420, 3516-3522, 3597-3601)
		struct Width { float scaled; bool thin; float width; };
		auto strokeWidth = [&](float width, uint32_t flags) {
			Width w;
			w.scaled = (flags & (1u << 5)) ? width : clampf(width * S().avgScale, 0.0f, 200.0f); // StrokeFlags::FixedWidth
			w.thin = w.scaled <= fringe;
			w.width = w.thin ? fringe : w.scaled;
			return w;
		};
		auto strokeFlagsOf = [&](uint32_t flags, bool aa, bool thin) { return VGX_STROKE_FLAGS((flags >> 2) & 3u, flags & 3u, aa, thin && aa); };
		auto localHandle = [&](uint16_t handle, uint16_t hflags, uint32_t first) { return (hflags & 0x0001u) ? (uint32_t)(uint16_t)(handle + first) : (uint32_t)handle; }; // isLocal, :4427
		switch (type) {
		case CT_BeginPath: beginPath(); break;
		case CT_MoveTo: if (!need(8)) { return VGX_E_INVALID_ARG; } pathArgs(VGX_CMD_MOVE_TO, f, 2); break;
		case CT_LineTo: if (!need(8)) { return VGX_E_INVALID_ARG; } pathArgs(VGX_CMD_LINE_TO, f, 2); break;
		case CT_CubicTo: if (!need(24)) { return VGX_E_INVALID_ARG; } pathArgs(VGX_CMD_CUBIC_TO, f, 6); break;
		case CT_QuadraticTo: if (!need(16)) { return VGX_E_INVALID_ARG; } pathArgs(VGX_CMD_QUAD_TO, f, 4); break;
		case CT_ArcTo: if (!need(20)) { return VGX_E_INVALID_ARG; } pathArgs(VGX_CMD_ARC_TO, f, 5); break;
		case CT_Arc: { // five floats + Winding::Enum (vg.cpp:2459-2472); vgx: sixth argument 1 = CW
			if (!need(24)) { return VGX_E_INVALID_ARG; }
			float a[6]; memcpy(a, d, 20); a[5] = u32at(20) == 1u ? 1.0f : 0.0f;
			pathArgs(VGX_CMD_ARC, a, 6);
		} break;
		case CT_Rect: if (!need(16)) { return VGX_E_INVALID_ARG; } pathArgs(VGX_CMD_RECT, f, 4); break;
		case CT_RoundedRect: if (!need(20)) { return VGX_E_INVALID_ARG; } pathArgs(VGX_CMD_ROUNDED_RECT, f, 5); break;
		case CT_RoundedRectVarying: if (!need(32)) { return VGX_E_INVALID_ARG; } pathArgs(VGX_CMD_ROUNDED_RECT_VARYING, f, 8); break;
		case CT_Circle: if (!need(12)) { return VGX_E_INVALID_ARG; } pathArgs(VGX_CMD_CIRCLE, f, 3); break;
		case CT_Ellipse: if (!need(16)) { return VGX_E_INVALID_ARG; } pathArgs(VGX_CMD_ELLIPSE, f, 4); break;
		case CT_ClosePath: pathArgs(VGX_CMD_CLOSE, f, 0); break;
		case CT_Polyline: { // uint32 numPoints + coordinates (vg.cpp:2551-2559)
			if (!need(4)) { return VGX_E_INVALID_ARG; }
			const uint32_t np = u32at(0);
			if (np == 0 || np > (psize - 4) / 8) { return VGX_E_INVALID_ARG; }
			pathArgs(VGX_CMD_POLYLINE, (const float*)(d + 4), np * 2);
		} break;

		case CT_FillPathColor: // uint32 flags, Color (vg.cpp:2619-2627); ctxFillPathColor :3061-3179
		case CT_FillPathImagePattern: { // uint32 flags, Color, uint16 handle, uint16 handle flags (:2640-2651); ctxFillPathImagePattern :3286-3399
			const bool img = type == CT_FillPathImagePattern;
			if (!need(img ? 12u : 8u)) { return VGX_E_INVALID_ARG; }
			const uint32_t flags = u32at(0), color = u32at(4);
			rawColor = color;
			const bool clip = recordClip && !img;
			const uint32_t col = clip ? kBlack : scaledColor(color, globalAlpha());
			if (!clip && !hasCache && (col >> 24) == 0) { break; } // transparent: the reference returns before transformPath
			if (!havePath) { ++nskipped; break; }
			latch();
			const bool aa = clip ? false : (flags & 0x04u) != 0;
			const uint32_t dt = clip ? (uint32_t)DT_Clip : (img ? (uint32_t)DT_ImagePattern : (uint32_t)DT_Textured);
			const uint32_t handle = clip ? 0xFFFFu : (img ? localHandle(u16at(8), u16at(10), firstImagePatternID) : (st0->font_image & 0xFFFFu)); // colour = Textured on the font atlas (createDrawCommand_VertexColor, vg.cpp:5210-5211)
			// PathType::Concave (:3133-3178): libtess2 stays with the caller -- a draw without a GPU mesh, see VGX_FILL_CONCAVE
			const uint32_t how = (flags & 0x01u) ? (VGX_FILL_CONCAVE | ((flags & 0x10u) ? VGX_FILL_EVEN_ODD : 0u)) : (uint32_t)VGX_FILL_ENABLE;
			emit(dt, handle, how | (aa ? VGX_FILL_AA : 0u), col, 0, 0, 0.0f);
		} break;
		case CT_FillPathGradient: { // uint32 flags, uint16 handle, uint16 handle flags (:2629-2638); ctxFillPathGradient :3181-3284
			if (!need(8)) { return VGX_E_INVALID_ARG; }
			const uint32_t flags = u32at(0);
			rawColor = 0;
			if (!havePath) { ++nskipped; break; }
			latch();
			const bool aa = (flags & 0x04u) != 0;
			// AA: strokerConvexFillAA(Colors::Black); else one colour = black with alpha 0xff * globalAlpha (:3212, 3228)
			const uint32_t col = aa ? kBlack : setAlpha(kBlack, (uint8_t)(0xff * S().alpha));
			const uint32_t how = (flags & 0x01u) ? (VGX_FILL_CONCAVE | ((flags & 0x10u) ? VGX_FILL_EVEN_ODD : 0u)) : (uint32_t)VGX_FILL_ENABLE; // :3245-3277
			emit(DT_ColorGradient, localHandle(u16at(4), u16at(6), firstGradientID), how | (aa ? VGX_FILL_AA : 0u), col, 0, 0, 0.0f);
		} break;

		case CT_StrokePathColor: { // float width, uint32 flags, Color (vg.cpp:2660-2669); ctxStrokePathColor :3401-3492
			if (!need(12)) { return VGX_E_INVALID_ARG; }
			const float width = f32at(0); const uint32_t flags = u32at(4), color = u32at(8);
			rawColor = color;
			const Width w = strokeWidth(width, flags);
			const float ga = globalAlpha();
			const float c = clampf(w.scaled, 0.0f, fringe);
			const float alphaScale = !w.thin ? ga : ga * (c * c);
			const uint32_t col = recordClip ? kBlack : scaledColor(color, alphaScale);
			if (!hasCache && (col >> 24) == 0) { break; }
			if (!havePath) { ++nskipped; break; }
			latch();
			const bool aa = recordClip ? false : (flags & 0x10u) != 0;
			emit(recordClip ? (uint32_t)DT_Clip : (uint32_t)DT_Textured, recordClip ? 0xFFFFu : (st0->font_image & 0xFFFFu), 0, 0, strokeFlagsOf(flags, aa, w.thin), col, w.width);
		} break;
		case CT_StrokePathGradient: { // float width, uint32 flags, uint16 handle, uint16 handle flags (:2671-2681); ctxStrokePathGradient :3494-3576
			if (!need(12)) { return VGX_E_INVALID_ARG; }
			const float width = f32at(0); const uint32_t flags = u32at(4);
			rawColor = 0;
			if (!havePath) { ++nskipped; break; }
			const bool aa = (flags & 0x10u) != 0;
			latch();
			const Width w = strokeWidth(width, flags);
			const uint32_t col = aa ? kBlack : setAlpha(kBlack, (uint8_t)(0xff * S().alpha)); // :3545-3546, 3550-3554
			emit(DT_ColorGradient, localHandle(u16at(8), u16at(10), firstGradientID), 0, 0, strokeFlagsOf(flags, aa, w.thin), col, w.width);
		} break;
		case CT_StrokePathImagePattern: { // float width, uint32 flags, Color, uint16 handle, uint16 handle flags (:2683-2695); ctxStrokePathImagePattern :3578-3668
			if (!need(16)) { return VGX_E_INVALID_ARG; }
			const float width = f32at(0); const uint32_t flags = u32at(4), color = u32at(8);
			rawColor = color;
			const Width w = strokeWidth(width, flags);
			const float ga = globalAlpha();
			const float c = clampf(w.scaled, 0.0f, fringe);
			const float alphaScale = w.thin ? ga : ga * (c * c); // sic (:3603): the Color flavour has the test the other way round
			const uint32_t col = scaledColor(color, alphaScale);
			if (!hasCache && (col >> 24) == 0) { break; }
			if (!havePath) { ++nskipped; break; }
			latch();
			const bool aa = (flags & 0x10u) != 0;
			emit(DT_ImagePattern, localHandle(u16at(12), u16at(14), firstImagePatternID), 0, 0, strokeFlagsOf(flags, aa, w.thin), col, w.width);
		} break;

		case CT_BeginClip: // ctxBeginClip, vg.cpp:3670-3683
			if (!need(4)) { return VGX_E_INVALID_ARG; }
			clipRule = u32at(0); clipFirst = drawBase + ndraws; clipNum = 0;
			recordClip = true; forceNew = true;
			break;
		case CT_EndClip: // :3685-3697. The region = the Clip draws among [clipFirst, clipFirst + clipNum): other draws may lie between them
			if (recordClip) { clipNum = drawBase + ndraws - clipFirst; }
			recordClip = false; forceNew = true; break;
		case CT_ResetClip: // :3699-3709
			if (clipFirst != 0xFFFFFFFFu) { clipFirst = 0xFFFFFFFFu; clipNum = 0; forceNew = true; }
			break;

		case CT_CreateLinearGradient: { // four floats, two colours (:2716-2733); ctxCreateLinearGradient :3711-3774
			if (!need(24)) { return VGX_E_INVALID_ARG; }
			if (nextGradient >= maxGradients) { break; }
			const uint32_t handle = nextGradient++;
			const float sx = f[0], sy = f[1], ex = f[2], ey = f[3];
			const float large = 1e5;
			float dx = ex - sx, dy = ey - sy;
			const float dd = vgm_sqrt(dx * dx + dy * dy);
			if (dd > 0.0001f) { dx /= dd; dy /= dd; } else { dx = 0; dy = 1; }
			const float g[6] = { dy, -dx, dx, dy, sx - dx * large, sy - dy * large };
			float pm[6], inv[6];
			mul3(S().m, g, pm); invert3(pm, inv);
			vgx_paint* pt = newPaint(DT_ColorGradient, handle, inv);
			pt->params[0] = large; pt->params[1] = large + dd * 0.5f; pt->params[2] = 0.0f; pt->params[3] = vgm_max(1.0f, dd);
			colors(pt, u32at(16), u32at(20));
		} break;
		case CT_CreateBoxGradient: { // six floats, two colours (:2735-2754); ctxCreateBoxGradient :3776-3826
			if (!need(32)) { return VGX_E_INVALID_ARG; }
			if (nextGradient >= maxGradients) { break; }
			const uint32_t handle = nextGradient++;
			const float g[6] = { 1.0f, 0.0f, 0.0f, 1.0f, f[0] + f[2] * 0.5f, f[1] + f[3] * 0.5f };
			float pm[6], inv[6];
			mul3(S().m, g, pm); invert3(pm, inv);
			vgx_paint* pt = newPaint(DT_ColorGradient, handle, inv);
			pt->params[0] = f[2] * 0.5f; pt->params[1] = f[3] * 0.5f; pt->params[2] = f[4]; pt->params[3] = vgm_max(1.0f, f[5]);
			colors(pt, u32at(24), u32at(28));
		} break;
		case CT_CreateRadialGradient: { // four floats, two colours (:2756-2773); ctxCreateRadialGradient :3828-3881
			if (!need(24)) { return VGX_E_INVALID_ARG; }
			if (nextGradient >= maxGradients) { break; }
			const uint32_t handle = nextGradient++;
			const float g[6] = { 1.0f, 0.0f, 0.0f, 1.0f, f[0], f[1] };
			float pm[6], inv[6];
			mul3(S().m, g, pm); invert3(pm, inv);
			const float r = (f[2] + f[3]) * 0.5f, fe = (f[3] - f[2]);
			vgx_paint* pt = newPaint(DT_ColorGradient, handle, inv);
			pt->params[0] = r; pt->params[1] = r; pt->params[2] = r; pt->params[3] = vgm_max(1.0f, fe);
			colors(pt, u32at(16), u32at(20));
		} break;
		case CT_CreateImagePattern: { // five floats + ImageHandle (:2775-2791); ctxCreateImagePattern :3883-3932
			if (!need(22)) { return VGX_E_INVALID_ARG; }
			const uint16_t image = u16at(20);
			if (image == 0xFFFFu || nextImagePattern >= maxImagePatterns) { break; }
			const uint32_t handle = nextImagePattern++;
			const float cs = vgm_cos(f[4]), sn = vgm_sin(f[4]);
			const float g[6] = { cs, sn, -sn, cs, f[0], f[1] };
			float pm[6], inv[6];
			mul3(S().m, g, pm); invert3(pm, inv);
			inv[0] /= f[2]; inv[1] /= f[3]; inv[2] /= f[2]; inv[3] /= f[3]; inv[4] /= f[2]; inv[5] /= f[3];
			vgx_paint* pt = newPaint(DT_ImagePattern, handle, inv);
			pt->image = image;
		} break;

		case CT_PushState: { const St top = stack.back(); stack.push_back(top); } break; // vg.cpp:3934-3943
		case CT_PopState: { // :3945-3966
			if (stack.size() <= 1) { return VGX_E_INVALID_ARG; }
			stack.pop_back();
			if (haveLastScissor) {
				const St& s = S();
				for (int i = 0; i < 4; ++i) { if (lastScissor[i] != (uint16_t)s.scissor[i]) { forceNew = true; } }
			}
			if (cullCmds) { skipCmds = (S().scissor[2] < 1.0f) || (S().scissor[3] < 1.0f); } // :4572-4576
		} break;
		case CT_ResetScissor: { // :3968-3976
			St& s = S();
			s.scissor[0] = s.scissor[1] = 0.0f; s.scissor[2] = st0->canvas_width; s.scissor[3] = st0->canvas_height;
			forceNew = true; skipCmds = false;
		} break;
		case CT_SetScissor: { // :3978-4000
			if (!need(16)) { return VGX_E_INVALID_ARG; }
			St& s = S();
			const float px = s.m[0] * f[0] + s.m[2] * f[1] + s.m[4], py = s.m[1] * f[0] + s.m[3] * f[1] + s.m[5]; // transformPos2D
			const float wx = s.m[0] * f[2] + s.m[2] * f[3], wy = s.m[1] * f[2] + s.m[3] * f[3];                   // transformVec2D
			const float cw = st0->canvas_width, ch = st0->canvas_height;
			const float minx = clampf(px, 0.0f, cw), miny = clampf(py, 0.0f, ch);
			const float maxx = clampf(px + wx, 0.0f, cw), maxy = clampf(py + wy, 0.0f, ch);
			s.scissor[0] = minx; s.scissor[1] = miny; s.scissor[2] = maxx - minx; s.scissor[3] = maxy - miny;
			forceNew = true;
			if (cullCmds) { skipCmds = (s.scissor[2] < 1.0f) || (s.scissor[3] < 1.0f); }
		} break;
		case CT_IntersectScissor: { // :4002-4030
			if (!need(16)) { return VGX_E_INVALID_ARG; }
			St& s = S();
			const float px = s.m[0] * f[0] + s.m[2] * f[1] + s.m[4], py = s.m[1] * f[0] + s.m[3] * f[1] + s.m[5];
			const float wx = s.m[0] * f[2] + s.m[2] * f[3], wy = s.m[1] * f[2] + s.m[3] * f[3];
			const float minx = vgm_max(px, s.scissor[0]), miny = vgm_max(py, s.scissor[1]);
			const float maxx = vgm_min(px + wx, s.scissor[0] + s.scissor[2]), maxy = vgm_min(py + wy, s.scissor[1] + s.scissor[3]);
			const float nw = vgm_max(0.0f, maxx - minx), nh = vgm_max(0.0f, maxy - miny);
			s.scissor[0] = minx; s.scissor[1] = miny; s.scissor[2] = nw; s.scissor[3] = nh;
			forceNew = true;
			if (cullCmds) { skipCmds = !(nw >= 1.0f && nh >= 1.0f); }
		} break;
		case CT_TransformIdentity: { St& s = S(); s.m[0] = 1; s.m[1] = 0; s.m[2] = 0; s.m[3] = 1; s.m[4] = 0; s.m[5] = 0; updateState(s); } break;
		case CT_TransformScale: { if (!need(8)) { return VGX_E_INVALID_ARG; } St& s = S(); s.m[0] = f[0] * s.m[0]; s.m[1] = f[0] * s.m[1]; s.m[2] = f[1] * s.m[2]; s.m[3] = f[1] * s.m[3]; updateState(s); } break; // :4044-4053
		case CT_TransformTranslate: { if (!need(8)) { return VGX_E_INVALID_ARG; } St& s = S(); s.m[4] += s.m[0] * f[0] + s.m[2] * f[1]; s.m[5] += s.m[1] * f[0] + s.m[3] * f[1]; updateState(s); } break; // :4055-4062
		case CT_TransformRotate: { // :4064-4082, bx::cos / bx::sin = the pinned ones of vgmath.h
			if (!need(4)) { return VGX_E_INVALID_ARG; }
			St& s = S();
			const float c = vgm_cos(f[0]), sn = vgm_sin(f[0]);
			float m[6];
			m[0] = c * s.m[0] + sn * s.m[2]; m[1] = c * s.m[1] + sn * s.m[3];
			m[2] = -sn * s.m[0] + c * s.m[2]; m[3] = -sn * s.m[1] + c * s.m[3];
			m[4] = s.m[4]; m[5] = s.m[5];
			memcpy(s.m, m, sizeof(m)); updateState(s);
		} break;
		case CT_TransformMult: { // six floats + TransformOrder::Enum (Pre = 0, Post = 1), :4084-4100
			if (!need(28)) { return VGX_E_INVALID_ARG; }
			St& s = S();
			float m[6], r[6]; memcpy(m, d, 24);
			if (u32at(24) == 1u) { mul3(s.m, m, r); } else { mul3(m, s.m, r); }
			memcpy(s.m, r, sizeof(r)); updateState(s);
		} break;
		case CT_SetViewBox: { // :4102-4121
			if (!need(16)) { return VGX_E_INVALID_ARG; }
			St& s = S();
			const float sxv = st0->canvas_width / f[2], syv = st0->canvas_height / f[3];
			s.m[0] = sxv * s.m[0]; s.m[1] = sxv * s.m[1]; s.m[2] = syv * s.m[2]; s.m[3] = syv * s.m[3];
			s.m[4] -= s.m[0] * f[0] + s.m[2] * f[1]; s.m[5] -= s.m[1] * f[0] + s.m[3] * f[1];
			updateState(s);
		} break;
		case CT_SetGlobalAlpha: if (!need(4)) { return VGX_E_INVALID_ARG; } S().alpha = f[0]; break;
		case CT_SubmitCommandList: { // uint16 handle (:2960-2967); :4611-4620
			if (!need(2)) { return VGX_E_INVALID_ARG; }
			const uint16_t h = u16at(0);
			if (!st0->lists || h >= st0->num_lists || (!st0->lists[h].bytes && st0->lists[h].size)) { ++nskipped; break; }
			const vgx_cmdlist_ref& L = st0->lists[h];
			if ((L.size % kAlign) != 0) { return VGX_E_INVALID_ARG; }
			const int rc = run((const uint8_t*)L.bytes, L.size, L.flags);
			if (rc != VGX_OK) { return rc; }
		} break;
		case CT_IndexedTriList: { // uint32 nv, float2 pos[nv], uint32 nuv, uv_t2 uv[nuv], uint32 nc, Color col[nc], uint32 ni, uint16 idx[ni], uint16 image
			// (clIndexedTriList vg.cpp:2566-2611; interpreter :4461-4477 -> ctxIndexedTriList :4129-4175)
			const uint32_t uvBytes = (st0->flags & VGX_CL_UV_FLOAT) ? 8u : 4u;
			uint64_t off = 0;
			if (!need(4)) { return VGX_E_INVALID_ARG; }
			const uint32_t nv = u32at(0); off = 4;
			if (nv > 65536u || off + (uint64_t)nv * 8 + 4 > psize) { return VGX_E_INVALID_ARG; }
			const uint32_t oPos = (uint32_t)off; off += (uint64_t)nv * 8;
			const uint32_t nuv = u32at((uint32_t)off); off += 4;
			if ((nuv != 0 && nuv != nv) || off + (uint64_t)nuv * uvBytes + 4 > psize) { return VGX_E_INVALID_ARG; }
			const uint32_t oUv = (uint32_t)off; off += (uint64_t)nuv * uvBytes;
			const uint32_t nc = u32at((uint32_t)off); off += 4;
			if ((nc != nv && nc != 1) || off + (uint64_t)nc * 4 + 4 > psize) { return VGX_E_INVALID_ARG; } // VG_CHECK(numColors == 1), :4163
			const uint32_t oCol = (uint32_t)off; off += (uint64_t)nc * 4;
			const uint32_t ni = u32at((uint32_t)off); off += 4;
			if (off + (uint64_t)ni * 2 + 2 > psize) { return VGX_E_INVALID_ARG; }
			const uint32_t oIdx = (uint32_t)off; off += (uint64_t)ni * 2;
			const uint16_t img = u16at((uint32_t)off);
			rawColor = 0;
			const uint32_t handle = img == 0xFFFFu ? st0->font_image : (uint32_t)img; // !isValid(img): the font atlas (:4131-4133)
			if (store) { // no tri_* arrays in the store pass = capacity 0: VGX_E_NOSPACE, never a silently dropped mesh
				if (!out->tri_meshes || !out->tri_pos || !out->tri_color || !out->tri_idx || ntriM >= out->cap_tri_meshes || (uint64_t)ntriV + nv > out->cap_tri_vertices || (uint64_t)ntriI + ni > out->cap_tri_indices) { overflow = true; }
				else {
					const float* m = S().m; // the state's transform at the command, not the path's latch
					for (uint32_t k = 0; k < nv; ++k) { // vgutil::batchTransformPositions, vg_util.cpp:266-272 (transformPos2D)
						const float x = f32at(oPos + 8 * k), y = f32at(oPos + 8 * k + 4);
						out->tri_pos[2 * (ntriV + k)] = m[0] * x + m[2] * y + m[4]; out->tri_pos[2 * (ntriV + k) + 1] = m[1] * x + m[3] * y + m[5];
						out->tri_color[ntriV + k] = u32at(oCol + (nc == nv ? 4 * k : 0));
					}
					if (out->tri_uv) {
						uint8_t* dst = (uint8_t*)out->tri_uv + (size_t)ntriV * uvBytes;
						if (nuv) { memcpy(dst, d + oUv, (size_t)nv * uvBytes); }
						else { for (uint32_t k = 0; k < nv; ++k) { memcpy(dst + (size_t)k * uvBytes, st0->white_uv, uvBytes); } }
					}
					if (ni) { memcpy(out->tri_idx + ntriI, d + oIdx, (size_t)ni * 2); }
					vgx_mesh& mr = out->tri_meshes[ntriM];
					mr.first_vertex = ntriV; mr.first_index = ntriI; mr.num_vertices = nv; mr.num_indices = ni;
					mr.draw = ndraws; mr.subpath_kind = (uint32_t)VGX_MESH_TRILIST << 28;
				}
			}
			++ntriM; ntriV += nv; ntriI += ni;
			if (!havePath) { beginPath(); } // the draw record needs SOME valid path index: an empty path when none is open
			emitHasMesh = true; // allocDrawCommand runs whatever the mesh holds (:4137): the PopState rule sees a draw command
			emit(DT_Textured, handle, VGX_FILL_TRILIST, 0, 0, 0, 0.0f);
			emitHasMesh = false;
			// the draw record carries the state transform (informative: the positions above are already transformed)
			if (store && ndraws - 1 < out->cap_draws) { memcpy(out->draws[ndraws - 1].mtx, S().m, sizeof(float) * 6); }
		} break;
		default: ++nskipped; break; // Text, TextBox
		}
	}
	--depth;
	return VGX_OK;
}

} // namespace

extern "C" int vgx_cmdlist_decode(const void* bytes, uint32_t size, const vgx_cmdlist_state* st0, vgx_cmdlist_out* out)
{
	if ((!bytes && size) || !st0 || !out || (size % kAlign) != 0 || ((uintptr_t)bytes & 3u) != 0) { // float operands are read in place: 4-byte aligned buffer
		return VGX_E_INVALID_ARG;
	}
	Decoder D;
	D.st0 = st0; D.out = out;
	D.store = out->cmd_type && out->cmd_arg_off && out->args && out->path_cmd_begin && out->draws; // else: count only
	D.overflow = false;
	D.npaths = D.ncmd = D.nargs = D.ndraws = D.nskipped = D.npaints = 0;
	D.havePath = false; D.transformed = false; D.pathScale = 1.0f;
	memset(D.pathMtx, 0, sizeof(D.pathMtx));
	if (D.store) { out->cmd_arg_off[0] = 0; out->path_cmd_begin[0] = 0; }
	D.stack.resize(1);
	St& s0 = D.stack[0];
	memcpy(s0.m, st0->mtx, sizeof(float) * 6);
	memcpy(s0.scissor, st0->scissor, sizeof(float) * 4);
	// all zero = "never set" (resetScissor) ONLY without VGX_CL_SCISSOR_SET: a decode chained from an earlier one of the frame
	// carries that decode's end_scissor, which may be a real empty rectangle (SetScissor(0,0,0,0), an intersection that came out empty)
	if (!(st0->flags & VGX_CL_SCISSOR_SET) && s0.scissor[0] == 0.0f && s0.scissor[1] == 0.0f && s0.scissor[2] == 0.0f && s0.scissor[3] == 0.0f) {
		s0.scissor[2] = st0->canvas_width; s0.scissor[3] = st0->canvas_height;
	}
	s0.alpha = st0->global_alpha;
	updateState(s0);
	D.forceNew = false; D.generation = st0->first_generation;
	D.haveLastScissor = st0->prev_cmd_valid != 0;
	memcpy(D.lastScissor, st0->prev_cmd_scissor, sizeof(D.lastScissor));
	D.drawBase = st0->draw_base;
	D.recordClip = st0->clip_recording != 0; D.clipRule = st0->clip_valid ? st0->clip_rule : 0u;
	D.clipFirst = st0->clip_valid ? st0->clip_first_draw : 0xFFFFFFFFu; D.clipNum = st0->clip_valid ? st0->clip_num_draws : 0u;
	D.nextGradient = st0->first_gradient; D.nextImagePattern = st0->first_image_pattern;
	D.maxGradients = st0->max_gradients ? st0->max_gradients : 64u;
	D.maxImagePatterns = st0->max_image_patterns ? st0->max_image_patterns : 64u;
	D.depth = 0; D.maxDepth = st0->max_depth ? st0->max_depth : 16u;

	const int rc = D.run((const uint8_t*)bytes, size, st0->flags);
	if (rc != VGX_OK) { return rc; }
	D.closePathRecord();
	out->num_paths = D.npaths; out->num_cmds = D.ncmd; out->num_args = D.nargs; out->num_draws = D.ndraws; out->num_paints = D.npaints;
	out->num_skipped = D.nskipped;
	out->num_tri_meshes = D.ntriM; out->num_tri_vertices = D.ntriV; out->num_tri_indices = D.ntriI;
	out->next_gradient = D.nextGradient; out->next_image_pattern = D.nextImagePattern;
	out->next_generation = D.generation + (D.forceNew ? 1u : 0u);
	memcpy(out->end_mtx, D.S().m, sizeof(float) * 6);
	out->end_global_alpha = D.S().alpha;
	out->end_clip_valid = D.clipFirst != 0xFFFFFFFFu ? 1u : 0u; out->end_clip_rule = D.clipRule;
	out->end_clip_first_draw = D.clipFirst != 0xFFFFFFFFu ? D.clipFirst : 0u; out->end_clip_num_draws = D.clipNum; out->end_clip_recording = D.recordClip ? 1u : 0u;
	memcpy(out->end_scissor, D.S().scissor, sizeof(float) * 4);
	out->reserved = 0;
	if (D.store && D.overflow) { return VGX_E_NOSPACE; }
	return VGX_OK;
}
