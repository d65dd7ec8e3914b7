"""Host-side path recording: the Python mirror of the reference's path-builder calls.

`PathSetBuilder` records vg::pathXXX-style calls (reference include/vg/path.h:24-35) into the SoA
command stream that include/vgx.h's vgx_pathset_desc describes. It only records; flattening
happens on the device (csrc/) or in the CPU oracle (oracle/).
"""
import numpy as np
from . import capi


class PathSetArrays:
    """Plain numpy arrays in vgx_pathset_desc layout."""

    def __init__(self, cmd_type, cmd_arg_off, args, path_cmd_begin):
        self.cmd_type = np.ascontiguousarray(cmd_type, dtype=np.uint8)
        self.cmd_arg_off = np.ascontiguousarray(cmd_arg_off, dtype=np.uint32)
        self.args = np.ascontiguousarray(args, dtype=np.float32)
        self.path_cmd_begin = np.ascontiguousarray(path_cmd_begin, dtype=np.uint32)
        assert self.cmd_arg_off.shape[0] == self.cmd_type.shape[0] + 1
        if self.args.shape[0] == 0:  # keep a valid pointer
            self.args = np.zeros(1, dtype=np.float32)

    @property
    def npaths(self):
        return int(self.path_cmd_begin.shape[0] - 1)

    @property
    def ncmd(self):
        return int(self.cmd_type.shape[0])

    def desc(self):
        d = capi.PathSetDesc()
        d.cmd_type = self.cmd_type.ctypes.data
        d.cmd_arg_off = self.cmd_arg_off.ctypes.data
        d.args = self.args.ctypes.data
        d.path_cmd_begin = self.path_cmd_begin.ctypes.data
        d.npaths = self.npaths
        d.ncmd = self.ncmd
        return d


def concat(sets):
    """One path set holding the paths of several, in order (draws of set k refer to it with their path index + the path
    counts of the sets in front of it): how successive vgx_cmdlist_decode results of one frame become one batch."""
    cmd_type, args, arg_off, pcb = [], [], [np.zeros(1, np.uint32)], [np.zeros(1, np.uint32)]
    nargs = ncmd = 0
    for ps in sets:
        na = int(ps.cmd_arg_off[-1])
        cmd_type.append(ps.cmd_type)
        args.append(ps.args[:na])
        arg_off.append(ps.cmd_arg_off[1:] + np.uint32(nargs))
        pcb.append(ps.path_cmd_begin[1:] + np.uint32(ncmd))
        nargs += na
        ncmd += ps.ncmd
    return PathSetArrays(np.concatenate(cmd_type) if cmd_type else np.zeros(0, np.uint8), np.concatenate(arg_off),
                         np.concatenate(args) if args else np.zeros(0, np.float32), np.concatenate(pcb))


class PathSetBuilder:
    def __init__(self):
        self._types = []
        self._args = []
        self._arg_off = [0]
        self._path_begin = [0]
        self._open = False

    # -- path bracketing (vg::beginPath; one Path object per draw in the reference) --
    def begin_path(self):
        if self._open:
            self.end_path()
        self._open = True
        return len(self._path_begin) - 1

    def end_path(self):
        self._path_begin.append(len(self._types))
        self._open = False

    def _cmd(self, t, *a):
        assert self._open, "begin_path() first"
        self._types.append(t)
        self._args.extend(float(x) for x in a)
        self._arg_off.append(len(self._args))

    # -- vg::pathXXX mirrors --
    def move_to(self, x, y): self._cmd(capi.CMD_MOVE_TO, x, y)
    def line_to(self, x, y): self._cmd(capi.CMD_LINE_TO, x, y)
    def cubic_to(self, c1x, c1y, c2x, c2y, x, y): self._cmd(capi.CMD_CUBIC_TO, c1x, c1y, c2x, c2y, x, y)
    def quadratic_to(self, cx, cy, x, y): self._cmd(capi.CMD_QUAD_TO, cx, cy, x, y)
    def close(self): self._cmd(capi.CMD_CLOSE)
    def arc_to(self, x1, y1, x2, y2, r): self._cmd(capi.CMD_ARC_TO, x1, y1, x2, y2, r)
    def arc(self, cx, cy, r, a0, a1, cw): self._cmd(capi.CMD_ARC, cx, cy, r, a0, a1, 1.0 if cw else 0.0)
    def rect(self, x, y, w, h): self._cmd(capi.CMD_RECT, x, y, w, h)
    def rounded_rect(self, x, y, w, h, r): self._cmd(capi.CMD_ROUNDED_RECT, x, y, w, h, r)
    def rounded_rect_varying(self, x, y, w, h, rtl, rtr, rbr, rbl): self._cmd(capi.CMD_ROUNDED_RECT_VARYING, x, y, w, h, rtl, rtr, rbr, rbl)
    def circle(self, cx, cy, r): self._cmd(capi.CMD_CIRCLE, cx, cy, r)
    def ellipse(self, cx, cy, rx, ry): self._cmd(capi.CMD_ELLIPSE, cx, cy, rx, ry)
    def polyline(self, coords): self._cmd(capi.CMD_POLYLINE, *np.asarray(coords, dtype=np.float32).reshape(-1))

    def arrays(self):
        if self._open:
            self.end_path()
        return PathSetArrays(np.array(self._types, dtype=np.uint8), np.array(self._arg_off, dtype=np.uint32),
                             np.array(self._args, dtype=np.float32), np.array(self._path_begin, dtype=np.uint32))


def make_draws(n):
    """ndarray of n vgx_draw records with the reference's defaults (createContext, vg.cpp:764-765:
    tessellation tolerance 0.25, fringe 1.0; identity transform, scale 1)."""
    d = np.zeros(n, dtype=capi.draw_dtype)
    d["scale"] = 1.0
    d["tess_tol"] = 0.25
    d["fringe"] = 1.0
    d["mtx"][:, 0] = 1.0
    d["mtx"][:, 3] = 1.0
    return d
