"""A frame as a list of vg calls that can be (a) played on the reference's Context, (b) recorded with the reference's own
vg::clXxx writers (both through oracle/pyvgref.py -> oracle/_ref/libvgref_vg.so = src/vg.cpp compiled unmodified) and
(c) written out with the test-side byte writer tests/cmdlist_util.py. TEST INFRASTRUCTURE."""
import numpy as np

import pyvgref as R


class Script:
    def __init__(self):
        self.ops = []

    def add(self, code, f=(), u=()):
        self.ops.append((code, tuple(float(x) for x in f), tuple(int(x) for x in u)))
        return self

    # path
    def begin_path(self): return self.add(R.BeginPath)
    def move_to(self, x, y): return self.add(R.MoveTo, (x, y))
    def line_to(self, x, y): return self.add(R.LineTo, (x, y))
    def cubic_to(self, *a): return self.add(R.CubicTo, a)
    def quadratic_to(self, *a): return self.add(R.QuadraticTo, a)
    def arc_to(self, *a): return self.add(R.ArcTo, a)
    def arc(self, cx, cy, r, a0, a1, cw): return self.add(R.Arc, (cx, cy, r, a0, a1), (1 if cw else 0,))
    def rect(self, *a): return self.add(R.Rect, a)
    def rounded_rect(self, *a): return self.add(R.RoundedRect, a)
    def rounded_rect_varying(self, *a): return self.add(R.RoundedRectVarying, a)
    def circle(self, *a): return self.add(R.Circle, a)
    def ellipse(self, *a): return self.add(R.Ellipse, a)
    def polyline(self, pts): pts = np.asarray(pts, np.float32).reshape(-1); return self.add(R.Polyline, pts, (len(pts) // 2,))
    def close_path(self): return self.add(R.ClosePath)
    # paint
    def fill(self, color, flags): return self.add(R.FillPathColor, (), (color, flags))
    def fill_gradient(self, handle, flags): return self.add(R.FillPathGradient, (), (handle & 0xFFFF, flags, handle >> 16))
    def fill_image(self, handle, color, flags): return self.add(R.FillPathImagePattern, (), (handle & 0xFFFF, flags, handle >> 16, color))
    def stroke(self, color, width, flags): return self.add(R.StrokePathColor, (width,), (color, flags))
    def stroke_gradient(self, handle, width, flags): return self.add(R.StrokePathGradient, (width,), (handle & 0xFFFF, flags, handle >> 16))
    def stroke_image(self, handle, color, width, flags): return self.add(R.StrokePathImagePattern, (width,), (handle & 0xFFFF, flags, handle >> 16, color))
    # Create*: the handle a command list hands out is LOCAL: idx counts per list from 0, flags = HandleFlags::LocalHandle (vg.cpp:2716-2791)
    def linear_gradient(self, sx, sy, ex, ey, icol, ocol): return self.add(R.CreateLinearGradient, (sx, sy, ex, ey), (icol, ocol))
    def box_gradient(self, x, y, w, h, r, f, icol, ocol): return self.add(R.CreateBoxGradient, (x, y, w, h, r, f), (icol, ocol))
    def radial_gradient(self, cx, cy, inr, outr, icol, ocol): return self.add(R.CreateRadialGradient, (cx, cy, inr, outr), (icol, ocol))
    def image_pattern(self, cx, cy, w, h, angle, image): return self.add(R.CreateImagePattern, (cx, cy, w, h, angle), (image,))
    # state
    def push(self): return self.add(R.PushState)
    def pop(self): return self.add(R.PopState)
    def identity(self): return self.add(R.TransformIdentity)
    def scale(self, x, y): return self.add(R.TransformScale, (x, y))
    def translate(self, x, y): return self.add(R.TransformTranslate, (x, y))
    def rotate(self, a): return self.add(R.TransformRotate, (a,))
    def mult(self, m, post): return self.add(R.TransformMult, m, (1 if post else 0,))
    def view_box(self, *a): return self.add(R.SetViewBox, a)
    def global_alpha(self, a): return self.add(R.SetGlobalAlpha, (a,))
    def reset_scissor(self): return self.add(R.ResetScissor)
    def set_scissor(self, *a): return self.add(R.SetScissor, a)
    def intersect_scissor(self, *a): return self.add(R.IntersectScissor, a)
    def begin_clip(self, rule): return self.add(R.BeginClip, (), (rule,))
    def end_clip(self): return self.add(R.EndClip)
    def reset_clip(self): return self.add(R.ResetClip)
    def indexed_tri_list(self, pos, colors, idx, uv=None, image=0xFFFF):
        """vg::indexedTriList / clIndexedTriList (vg.h:476, 509). uv: None or an (nv, 2) array of the build's uv_t (int16 / float32)."""
        pos = np.asarray(pos, np.float32).reshape(-1, 2)
        colors = np.asarray(colors, np.uint32).reshape(-1)
        idx = np.asarray(idx, np.uint16).reshape(-1)
        words = [np.asarray([pos.shape[0], 0 if uv is None else 1, colors.shape[0], idx.shape[0], image], np.uint32), colors]
        if uv is not None:
            uv = np.ascontiguousarray(uv)
            assert uv.shape == (pos.shape[0], 2) and uv.dtype in (np.int16, np.float32)
            words.append(uv.reshape(-1).view(np.uint32))
        packed = np.zeros((idx.shape[0] + 1) // 2 * 2, np.uint16)
        packed[:idx.shape[0]] = idx
        words.append(packed.view(np.uint32))
        self.ops.append((R.IndexedTriList, pos.reshape(-1).copy(), np.concatenate(words)))
        return self

    def submit(self, child): return self.add(R.SubmitCommandList, (), (child,))

    def play(self, rc, cl):
        """Issue every call on the reference (cl = R.IMMEDIATE: vg::xxx on the Context; else vg::clXxx into that list).
        Returns the values the Create* calls returned."""
        ret = []
        for code, f, u in self.ops:
            v = rc.op(cl, code, f, u)
            if R.CreateLinearGradient <= code <= R.CreateImagePattern:
                ret.append(v)
        return ret


LOCAL = 1 << 16  # HandleFlags::LocalHandle in the flags half of a VG_HANDLE32 as vgr_op returns it (idx | flags << 16)


def add_path(s, ps, p):
    """Append path p of a PathSetArrays to the script (BeginPath + its commands)."""
    s.begin_path()
    for k in range(int(ps.path_cmd_begin[p]), int(ps.path_cmd_begin[p + 1])):
        t = int(ps.cmd_type[k])
        a = ps.args[int(ps.cmd_arg_off[k]):int(ps.cmd_arg_off[k + 1])].tolist()
        if t == 6:
            s.arc(a[0], a[1], a[2], a[3], a[4], a[5] != 0)
        elif t == 12:
            s.polyline(a)
        else:
            s.add(R._PATH_OPS[t], a)
    return s
