// vgo_driver.inl -- batch driver shared by the two CPU oracles (TEST INFRASTRUCTURE, not product).
//
// Included by oracle/ref_capi.cpp  with VGO_ENGINE = vg   (the reference's own compiled sources) and
// by          oracle/port_capi.cpp with VGO_ENGINE = vgo  (the restatement in oracle/vgo_port.cpp).
// VGO_XFORM names the scalar batchTransformPositions of that engine.
//
// For every draw it does exactly what the reference's callers do around the hot path:
//   ctxBeginPath     -> pathReset + strokerReset                      (src/vg.cpp:2969-2981)
//   ctxMoveTo/...    -> pathXXX                                       (src/vg.cpp:2983-3059)
//   transformPath    -> vgutil::batchTransformPositions               (src/vg.cpp:4957-4975)
//   ctxFillPathColor -> strokerConvexFill[AA] per sub-path >= 3 verts (src/vg.cpp:3099-3131)
//   ctxStrokePath... -> strokerPolylineStroke[AA|AAThin] per sub-path >= 2 verts (src/vg.cpp:3448-3485)
// and appends the returned vg::Mesh to contiguous streams, the way createDrawCommand_VertexColor
// copies them (src/vg.cpp:5207-5244). Output layout is the one documented in include/vgx.h.

#include "../include/vgx.h"
#include <vector>
#include <string.h>

namespace {

struct VgoArgCount { int n; };
static const int kArgCount[VGX_CMD_COUNT_] = { 2, 2, 6, 4, 0, 5, 6, 4, 5, 8, 3, 4, -1 };

struct VgoEngineState
{
	bx::ShimAllocator alloc;
	VGO_ENGINE::Path* path;
	VGO_ENGINE::Stroker* stroker;
	std::vector<float> xf;
	VgoEngineState() { path = VGO_ENGINE::createPath(&alloc); stroker = VGO_ENGINE::createStroker(&alloc); }
	~VgoEngineState() { VGO_ENGINE::destroyStroker(stroker); VGO_ENGINE::destroyPath(path); }
};

#ifndef VGO_REFERENCE_IS_SSE
#define VGO_REFERENCE_IS_SSE 0 /* 1 in oracle/_ref/libvgref_sse.so: strokerConvexFillAA is the reference's SSE2 variant */
#endif
// Index stream of the SSE2 strokerConvexFillAA (stroker.cpp:610-701) for a polygon of n corners: the first fringe quad (:616-618),
// per fan triangle t the triangle (0, s, s + 2) followed by the fringe quad of the next edge (s, s + 1, s + 3, s, s + 3, s + 2) with
// s = 2 t + 2 (:622-690: four at a time from delta tables, then the remainder), the wrap-around quad (:693-699). 9 n - 6 indices.
static void vgo_fill_aa_sse_order(uint32_t n, std::vector<uint16_t>& out)
{
	out.clear();
	const uint16_t q0[6] = { 0, 1, 3, 0, 3, 2 };
	out.insert(out.end(), q0, q0 + 6);
	uint32_t s = 2;
	for (uint32_t t = 0; t + 2 < n; ++t, s += 2) {
		const uint32_t g[9] = { 0, s, s + 2, s, s + 1, s + 3, s, s + 3, s + 2 };
		for (int k = 0; k < 9; ++k) { out.push_back((uint16_t)g[k]); }
	}
	const uint32_t last = (n - 1) << 1;
	const uint32_t ql[6] = { last, last + 1, 1, last, 1, 0 };
	for (int k = 0; k < 6; ++k) { out.push_back((uint16_t)ql[k]); }
}

static void vgoReplay(VGO_ENGINE::Path* path, const vgx_pathset_desc* ps, uint32_t pathID)
{
	using namespace VGO_ENGINE;
	const uint32_t c0 = ps->path_cmd_begin[pathID];
	const uint32_t c1 = ps->path_cmd_begin[pathID + 1];
	for (uint32_t c = c0; c < c1; ++c) {
		const float* a = &ps->args[ps->cmd_arg_off[c]];
		const uint32_t na = ps->cmd_arg_off[c + 1] - ps->cmd_arg_off[c];
		switch (ps->cmd_type[c]) {
		case VGX_CMD_MOVE_TO: pathMoveTo(path, a[0], a[1]); break;
		case VGX_CMD_LINE_TO: pathLineTo(path, a[0], a[1]); break;
		case VGX_CMD_CUBIC_TO: pathCubicTo(path, a[0], a[1], a[2], a[3], a[4], a[5]); break;
		case VGX_CMD_QUAD_TO: pathQuadraticTo(path, a[0], a[1], a[2], a[3]); break;
		case VGX_CMD_CLOSE: pathClose(path); break;
		case VGX_CMD_ARC_TO: pathArcTo(path, a[0], a[1], a[2], a[3], a[4]); break;
		case VGX_CMD_ARC: pathArc(path, a[0], a[1], a[2], a[3], a[4], a[5] != 0.0f ? Winding::CW : Winding::CCW); break;
		case VGX_CMD_RECT: pathRect(path, a[0], a[1], a[2], a[3]); break;
		case VGX_CMD_ROUNDED_RECT: pathRoundedRect(path, a[0], a[1], a[2], a[3], a[4]); break;
		case VGX_CMD_ROUNDED_RECT_VARYING: pathRoundedRectVarying(path, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]); break;
		case VGX_CMD_CIRCLE: pathCircle(path, a[0], a[1], a[2]); break;
		case VGX_CMD_ELLIPSE: pathEllipse(path, a[0], a[1], a[2], a[3]); break;
		case VGX_CMD_POLYLINE: pathPolyline(path, a, na / 2); break;
		default: break;
		}
	}
}

struct VgoSink
{
	vgx_mesh_out* out;
	vgx_sizes* sizes;
	bool overflow;

	void add(const VGO_ENGINE::Mesh& m, uint32_t uniformColor, uint32_t draw, uint32_t subpath, uint32_t kind, uint32_t polyN)
	{
		sizes->num_elements += polyN;
		if (kind == VGX_MESH_FILL || kind == VGX_MESH_FILL_AA) { sizes->num_fill_elements += polyN; }
		const uint64_t v0 = sizes->num_vertices;
		const uint64_t i0 = sizes->num_indices;
		const uint64_t m0 = sizes->num_meshes;
		sizes->num_vertices += m.m_NumVertices;
		sizes->num_indices += m.m_NumIndices;
		sizes->num_meshes += 1;
		if (!out) {
			return;
		}
		if (sizes->num_vertices > out->cap_vertices || sizes->num_indices > out->cap_indices || sizes->num_meshes > out->cap_meshes) {
			overflow = true;
			return;
		}
		if (out->pos) { memcpy(out->pos + v0 * 2, m.m_PosBuffer, sizeof(float) * 2 * m.m_NumVertices); }
		if (out->color) {
			if (m.m_ColorBuffer) {
				memcpy(out->color + v0, m.m_ColorBuffer, sizeof(uint32_t) * m.m_NumVertices);
			} else {
				for (uint32_t i = 0; i < m.m_NumVertices; ++i) { out->color[v0 + i] = uniformColor; }
			}
		}
		if (out->idx) { memcpy(out->idx + i0, m.m_IndexBuffer, sizeof(uint16_t) * m.m_NumIndices); }
		if (out->meshes) {
			vgx_mesh& r = out->meshes[m0];
			r.first_vertex = v0;
			r.first_index = i0;
			r.num_vertices = m.m_NumVertices;
			r.num_indices = m.m_NumIndices;
			r.draw = draw;
			r.subpath_kind = (subpath & 0x0FFFFFFFu) | (kind << 28);
		}
	}
};

static int vgoRun(const vgx_pathset_desc* ps, const vgx_draw* draws, uint64_t ndraws, int applyTransform, const vgx_flat_out* flat, vgx_mesh_out* meshOut, bool tessellate, vgx_sizes* sizes)
{
	using namespace VGO_ENGINE;
	if (!ps || (!draws && ndraws) || !sizes) {
		return VGX_E_INVALID_ARG;
	}
	memset(sizes, 0, sizeof(*sizes));
	VgoEngineState st;
	VgoSink sink = { meshOut, sizes, false };
	bool flatOverflow = false;

	for (uint64_t d = 0; d < ndraws; ++d) {
		const vgx_draw& dr = draws[d];
		if (dr.path >= ps->npaths) {
			return VGX_E_INVALID_ARG;
		}
		pathReset(st.path, dr.scale, dr.tess_tol);
		strokerReset(st.stroker, dr.scale, dr.tess_tol, dr.fringe);
		vgoReplay(st.path, ps, dr.path);

		const uint32_t nv = pathGetNumVertices(st.path);
		const uint32_t nsp = pathGetNumSubPaths(st.path);
		const float* verts = pathGetVertices(st.path);
		const SubPath* sp = pathGetSubPaths(st.path);
		if (st.xf.size() < (size_t)nv * 2 + 2) { st.xf.resize((size_t)nv * 2 + 2); }
		if (nv) { VGO_XFORM(verts, nv, st.xf.data(), dr.mtx); }

		const uint64_t pv0 = sizes->num_poly_vertices;
		const uint64_t sp0 = sizes->num_subpaths;
		const uint64_t mesh0 = sizes->num_meshes;
		sizes->num_poly_vertices += nv;
		sizes->num_subpaths += nsp;
		sizes->num_cmd_instances += ps->path_cmd_begin[dr.path + 1] - ps->path_cmd_begin[dr.path];

		if (flat) {
			if (sizes->num_poly_vertices > flat->cap_poly_vertices || sizes->num_subpaths > flat->cap_subpaths) {
				flatOverflow = true;
			} else {
				if (flat->poly && nv) { memcpy(flat->poly + pv0 * 2, applyTransform ? st.xf.data() : verts, sizeof(float) * 2 * nv); }
				if (flat->subpaths) {
					for (uint32_t i = 0; i < nsp; ++i) {
						vgx_subpath& o = flat->subpaths[sp0 + i];
						o.first_vertex = pv0 + sp[i].m_FirstVertexID;
						o.num_vertices = sp[i].m_NumVertices;
						o.flags = sp[i].m_IsClosed ? 1u : 0u;
					}
				}
			}
		}

		if (tessellate) {
			const float* tv = st.xf.data();
			if (dr.fill_flags & VGX_FILL_ENABLE) {
				for (uint32_t i = 0; i < nsp; ++i) {
					if (sp[i].m_NumVertices < 3) { continue; }
					Mesh mesh;
					const float* vtx = &tv[sp[i].m_FirstVertexID << 1];
					if (dr.fill_flags & VGX_FILL_AA) {
						strokerConvexFillAA(st.stroker, &mesh, vtx, sp[i].m_NumVertices, dr.fill_color);
#if !VGO_REFERENCE_IS_SSE
						// VGX_FILL_INDEX_ORDER_SSE: the index order of the reference's SSE2 variant, restated (the values depend on
						// the vertex count only). In the SSE build of oracle/_ref the reference writes this order by itself --
						// tests/test_oracle_golden.py compares the two.
						std::vector<uint16_t> sseIdx;
						if (dr.fill_flags & VGX_FILL_INDEX_ORDER_SSE) {
							vgo_fill_aa_sse_order(sp[i].m_NumVertices, sseIdx);
							mesh.m_IndexBuffer = sseIdx.data();
						}
#endif
						sink.add(mesh, dr.fill_color, (uint32_t)d, i, VGX_MESH_FILL_AA, sp[i].m_NumVertices);
					} else {
						strokerConvexFill(st.stroker, &mesh, vtx, sp[i].m_NumVertices);
						sink.add(mesh, dr.fill_color, (uint32_t)d, i, VGX_MESH_FILL, sp[i].m_NumVertices);
					}
				}
			}
			if (dr.stroke_flags & VGX_STROKE_ENABLE) {
				const LineCap::Enum cap = (LineCap::Enum)VGX_STROKE_CAP(dr.stroke_flags);
				const LineJoin::Enum join = (LineJoin::Enum)VGX_STROKE_JOIN(dr.stroke_flags);
				if ((uint32_t)cap > 2 || (uint32_t)join > 2) {
					return VGX_E_INVALID_ARG;
				}
				for (uint32_t i = 0; i < nsp; ++i) {
					if (sp[i].m_NumVertices < 2) { continue; }
					Mesh mesh;
					const float* vtx = &tv[sp[i].m_FirstVertexID << 1];
					const bool closed = sp[i].m_IsClosed;
					if (dr.stroke_flags & VGX_STROKE_AA) {
						if (dr.stroke_flags & VGX_STROKE_THIN) {
							strokerPolylineStrokeAAThin(st.stroker, &mesh, vtx, sp[i].m_NumVertices, closed, dr.stroke_color, cap, join);
							sink.add(mesh, dr.stroke_color, (uint32_t)d, i, VGX_MESH_STROKE_AA_THIN, sp[i].m_NumVertices);
						} else {
							strokerPolylineStrokeAA(st.stroker, &mesh, vtx, sp[i].m_NumVertices, closed, dr.stroke_color, dr.stroke_width, cap, join);
							sink.add(mesh, dr.stroke_color, (uint32_t)d, i, VGX_MESH_STROKE_AA, sp[i].m_NumVertices);
						}
					} else {
						strokerPolylineStroke(st.stroker, &mesh, vtx, sp[i].m_NumVertices, closed, dr.stroke_width, cap, join);
						sink.add(mesh, dr.stroke_color, (uint32_t)d, i, VGX_MESH_STROKE, sp[i].m_NumVertices);
					}
				}
			}
		}

		if (flat && flat->draw_info && !flatOverflow) {
			vgx_draw_info& di = flat->draw_info[d];
			di.first_poly_vertex = pv0;
			di.first_subpath = sp0;
			di.first_mesh = mesh0;
			di.num_poly_vertices = nv;
			di.num_subpaths = nsp;
			di.num_meshes = (uint32_t)(sizes->num_meshes - mesh0);
			di.flags = 0;
		}
	}
	return (flatOverflow || sink.overflow) ? VGX_E_NOSPACE : VGX_OK;
}

} // namespace

extern "C" {

// Flatten only. flat == NULL -> count only.
int vgo_flatten(const vgx_pathset_desc* ps, const vgx_draw* draws, uint64_t ndraws, int applyTransform, const vgx_flat_out* flat, vgx_sizes* sizes)
{
	return vgoRun(ps, draws, ndraws, applyTransform, flat, nullptr, false, sizes);
}

// Flatten + transform + stroker. out == NULL -> count only. flat (optional) also receives the
// transformed polyline and sub-path tables.
int vgo_tessellate(const vgx_pathset_desc* ps, const vgx_draw* draws, uint64_t ndraws, const vgx_flat_out* flat, vgx_mesh_out* out, vgx_sizes* sizes)
{
	return vgoRun(ps, draws, ndraws, 1, flat, out, true, sizes);
}

// Draw-command assembly of a frame of meshes that all share one draw state: CPU statement of what
// src/vg.cpp does per mesh in createDrawCommand_VertexColor (:5207-5244) via allocVertices (:5321-5342), allocIndices
// (:5344-5357) and allocDrawCommand (:5359-5407), starting from an empty first vertex buffer and an empty index
// buffer:
//   allocVertices     if (vb.count + numVertices > maxVBVertices) { new vertex buffer; forceNewDrawCommand = true; }
//                     firstVertexID = vb.count; vb.count += numVertices;
//   allocIndices      firstIndexID = ib.count; ib.count += numIndices;       (one index buffer per frame)
//   allocDrawCommand  if (!forceNew && a previous command exists && same type/handle) reuse it, else create
//                     {vb id, firstVertexID, firstIndexID, numVertices = 0, numIndices = 0}; forceNew = false;
//   createDrawCommand dstIndex = ib + cmd.firstIndexID + cmd.numIndices; rebase(indices, numIndices, dstIndex,
//                     (uint16_t)cmd.numVertices); cmd.numVertices += numVertices; cmd.numIndices += numIndices;
// PARITY UNPINNED for this bookkeeping: vg.cpp cannot be compiled without bgfx and the reference has no tests; only the
// rebase primitive is the reference's own code in the `reference` oracle (VGO_REBASE = vgutil::batchTransformDrawIndices,
// vg_util.cpp:447-520). idx_in holds mesh-local indices (vgo_tessellate's idx stream), idx_out receives the index buffer.
// mesh_key (may be NULL = one state): what allocDrawCommand compares before merging into the previous command (type and
// handle, vg.cpp:5376-5379), one word per mesh; a change starts a new command inside the same vertex buffer.
int vgo_assemble(const vgx_mesh* meshes, uint64_t nmeshes, const uint16_t* idx_in, uint16_t* idx_out, uint32_t maxVBVertices,
	vgx_drawcmd* cmds, uint64_t capCmds, uint64_t* numCmds, const uint32_t* mesh_key)
{
	if (maxVBVertices == 0) { maxVBVertices = 65536u; }
	uint32_t vbCount = 0, vbID = 0;      // current vertex buffer (vg.cpp:5326)
	uint64_t ibCount = 0;                // IndexBuffer::m_Count
	uint64_t vbGlobalStart = 0;          // where the current vertex buffer starts in the concatenated vertex streams
	bool forceNew = false;
	uint64_t ncmd = 0;
	vgx_drawcmd cur = {};
	bool have = false;
	int status = VGX_OK;
	for (uint64_t m = 0; m < nmeshes; ++m) {
		const uint32_t nv = meshes[m].num_vertices, ni = meshes[m].num_indices;
		if (nv > maxVBVertices) { status = VGX_E_MESH_TOO_LARGE; } // VG_CHECK(numVertices < m_MaxVBVertices), vg.cpp:5323
		// allocVertices
		if (vbCount + (uint64_t)nv > maxVBVertices) {
			vbGlobalStart += vbCount;
			++vbID;
			vbCount = 0;
			forceNew = true;
		}
		const uint32_t firstVertexID = vbCount;
		vbCount += nv;
		// allocIndices
		const uint64_t firstIndexID = ibCount;
		ibCount += ni;
		// allocDrawCommand: merge only when nothing forced a new command and type / handle agree (vg.cpp:5368-5381)
		const uint32_t key = mesh_key ? mesh_key[m] : 0u;
		if (forceNew || !have || cur.state_key != key) {
			if (have) {
				if (ncmd < capCmds) { cmds[ncmd] = cur; }
				++ncmd;
			}
			cur.vertex_buffer = vbID;
			cur.first_vertex = vbGlobalStart + firstVertexID; // m_FirstVertexID is firstVertexID (0 for a new buffer)
			cur.first_index = firstIndexID;
			cur.first_mesh = m;
			cur.first_vertex_in_vb = firstVertexID;
			cur.state_key = key;
			cur.num_vertices = 0; cur.num_indices = 0; cur.num_meshes = 0;
			have = true;
			forceNew = false;
		}
		// createDrawCommand_VertexColor: index rebase into the frame's index buffer
		if (idx_in && idx_out) {
			VGO_REBASE(idx_in + meshes[m].first_index, ni, idx_out + cur.first_index + cur.num_indices, (uint16_t)cur.num_vertices);
		}
		cur.num_vertices += nv;
		cur.num_indices += ni;
		cur.num_meshes += 1;
	}
	if (have) {
		if (ncmd < capCmds) { cmds[ncmd] = cur; }
		++ncmd;
	}
	*numCmds = ncmd;
	if (ncmd > capCmds && status == VGX_OK) { status = VGX_E_NOSPACE; }
	return status;
}

// Shape cache. vgo_cache_localize = addCachedCommand (src/vg.cpp:5808-5841): every mesh's positions times
// the inverse of the state transform its draw was recorded under (beginCachedCommand :5773-5790 takes the inverse with
// vgutil::invertMatrix3). vgo_cache_submit = submitCachedMesh(Color) (vg.cpp:6137-6166) for a list of instances:
// batchTransformPositions with the instance transform, then what createDrawCommand_VertexColor copies (positions,
// colours, indices) appended mesh after mesh. The arithmetic (VGO_INVERT3 / VGO_XFORM) is the reference's own
// vg_util.cpp code in the `reference` oracle; the loops around it are restated (vg.cpp cannot be built without bgfx).
int vgo_cache_localize(const vgx_draw* draws, uint64_t ndraws, float* pos, const vgx_mesh* meshes, uint64_t nmeshes)
{
	for (uint64_t m = 0; m < nmeshes; ++m) {
		if (meshes[m].draw >= ndraws) { return VGX_E_INVALID_ARG; }
		float inv[6];
		VGO_INVERT3(draws[meshes[m].draw].mtx, inv);
		float* p = pos + 2 * meshes[m].first_vertex;
		std::vector<float> tmp(p, p + 2 * (size_t)meshes[m].num_vertices);
		VGO_XFORM(tmp.data(), meshes[m].num_vertices, p, inv);
	}
	return VGX_OK;
}

int vgo_cache_submit(const vgx_cache_desc* cache, const vgx_cache_instance* inst, uint64_t ninst, const vgx_mesh_out* out, vgx_sizes* sizes)
{
	memset(sizes, 0, sizeof(*sizes));
	uint64_t nv = 0, ni = 0, nm = 0;
	bool overflow = false;
	for (uint64_t i = 0; i < ninst; ++i) {
		if (inst[i].first_mesh > cache->num_meshes || inst[i].num_meshes > cache->num_meshes - inst[i].first_mesh) { return VGX_E_INVALID_ARG; }
		for (uint64_t k = 0; k < inst[i].num_meshes; ++k) {
			const vgx_mesh& src = cache->meshes[inst[i].first_mesh + k];
			if (out) {
				if (nv + src.num_vertices > out->cap_vertices || ni + src.num_indices > out->cap_indices || (out->meshes && nm + 1 > out->cap_meshes)) {
					overflow = true;
				} else {
					VGO_XFORM(cache->pos + 2 * src.first_vertex, src.num_vertices, out->pos + 2 * nv, inst[i].mtx);
					const uint32_t kind = src.subpath_kind >> 28;
					if (kind == VGX_MESH_FILL || kind == VGX_MESH_STROKE) { // cached without colours (numColors == 1, vg.cpp:5826-5834): the replaying command's colour (:6159-6160)
						for (uint32_t v = 0; v < src.num_vertices; ++v) { out->color[nv + v] = inst[i].color; }
					} else {
						memcpy(out->color + nv, cache->color + src.first_vertex, sizeof(uint32_t) * src.num_vertices);
					}
					memcpy(out->idx + ni, cache->idx + src.first_index, sizeof(uint16_t) * src.num_indices);
					if (out->meshes) {
						vgx_mesh r = src;
						r.first_vertex = nv; r.first_index = ni; r.draw = (uint32_t)i;
						out->meshes[nm] = r;
					}
				}
			}
			nv += src.num_vertices; ni += src.num_indices; nm += 1;
		}
	}
	sizes->num_vertices = nv; sizes->num_indices = ni; sizes->num_meshes = nm;
	return overflow ? VGX_E_NOSPACE : VGX_OK;
}

const char* vgo_engine_name(void) { return VGO_ENGINE_NAME; }

} // extern "C"
