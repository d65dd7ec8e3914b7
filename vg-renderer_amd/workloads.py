"""Seeded synthetic workloads for BASELINE.json's configs.

All generators are pure numpy on np.random.RandomState (frozen stream => identical inputs in this
container and on the GPU box) and return (PathSetArrays, draws ndarray[draw_dtype]).

config 0  single_cubic()            one cubic -> 16-segment polyline, Butt/Miter AA stroke
config 1  random_cubics(1M)         independent cubics, flatten only
config 2  tiger(K)                  "tiger-like" 240-path drawing x K instances, convexFillAA + strokeAA
config 3  random_walk_polylines()   10k polylines x 1k segments, Round caps + Round joins
config 4  tiger(80k) sharded 8 ways (see dist.py)

The Ghostscript tiger itself is not in the reference (only a screenshot, img/vgrenderer_tiger.png) and
there is no network, so config 2 uses a seeded generator with the tiger's structure: 240 paths, closed
smooth cubic sub-paths of very different sizes, every path filled, about a third also stroked with a
mix of hairline (Thin) and regular widths.
"""
import numpy as np
from . import capi
from .pathset import PathSetBuilder, make_draws


def _color(r, g, b, a=255):
    return (a << 24) | (b << 16) | (g << 8) | r


def set_fill(draws, sel, color, aa=True):
    draws["fill_flags"][sel] = capi.fill_flags(aa)
    draws["fill_color"][sel] = color


def set_stroke(draws, sel, color, width, cap=capi.CAP_BUTT, join=capi.JOIN_MITER, aa=True, avg_scale=1.0,
               fringe=1.0, global_alpha=1.0, fixed_width=False):
    """Mirror of the caller logic in ctxStrokePathColor (reference src/vg.cpp:3416-3433): scales and
    clamps the width, switches to the Thin stroker when width <= fringe and scales alpha by width^2."""
    sw = np.float32(width) if fixed_width else np.float32(min(max(np.float32(width) * np.float32(avg_scale), 0.0), 200.0))
    thin = bool(sw <= np.float32(fringe))
    alpha_scale = np.float32(global_alpha)
    if thin:
        c = np.float32(min(max(sw, np.float32(0.0)), np.float32(fringe)))
        alpha_scale = np.float32(global_alpha) * (c * c)
    a = (color >> 24) & 0xFF
    col = (color & 0x00FFFFFF) | (int(np.uint8(np.float32(alpha_scale) * np.float32(a))) << 24)
    draws["stroke_flags"][sel] = capi.stroke_flags(cap, join, aa, thin and aa)
    draws["stroke_color"][sel] = col
    draws["stroke_width"][sel] = np.float32(fringe) if thin else sw
    return thin


# ---- config 0 ----------------------------------------------------------------------------------
def single_cubic():
    b = PathSetBuilder()
    b.begin_path()
    b.move_to(0, 0)
    b.cubic_to(22.5, 0, 45, 22.5, 45, 45)
    b.end_path()
    d = make_draws(1)
    set_stroke(d, 0, 0xFF0000FF, 10.0)
    return b.arrays(), d


# ---- config 1 ----------------------------------------------------------------------------------
def random_cubics(n, seed=1234, box=1000.0):
    """n independent paths, each moveTo + cubicTo with 8 coordinates uniform in [0, box).
    Portable mapping: 24 random bits * 2^-24 * box (SURVEY 8d, config 2 note)."""
    rs = np.random.RandomState(seed)
    bits = rs.randint(0, 1 << 24, size=(n, 8)).astype(np.float32)
    pts = bits * np.float32(2.0 ** -24) * np.float32(box)
    cmd_type = np.tile(np.array([capi.CMD_MOVE_TO, capi.CMD_CUBIC_TO], dtype=np.uint8), n)
    arg_off = np.zeros(2 * n + 1, dtype=np.uint32)
    arg_off[1::2] = np.arange(n, dtype=np.uint32) * 8 + 2
    arg_off[2::2] = np.arange(1, n + 1, dtype=np.uint32) * 8
    path_begin = np.arange(n + 1, dtype=np.uint32) * 2
    from .pathset import PathSetArrays
    ps = PathSetArrays(cmd_type, arg_off, pts.reshape(-1), path_begin)
    d = make_draws(n)
    d["path"] = np.arange(n, dtype=np.uint32)
    return ps, d


# ---- config 2 ----------------------------------------------------------------------------------
TIGER_PALETTE = [_color(*c) for c in [
    (255, 255, 255), (0, 0, 0), (204, 114, 38), (233, 127, 58), (242, 204, 153), (229, 102, 140),
    (178, 52, 41), (165, 38, 12), (255, 114, 127), (101, 153, 0), (153, 204, 50), (76, 0, 0),
    (153, 38, 0), (234, 142, 81), (76, 76, 76), (204, 204, 204)]]


def tiger_paths(seed=2024, npaths=240, closed=True):
    """Seeded tiger-like drawing: returns (PathSetArrays, per-path op table).
    ops[p] = dict(fill_color, stroke (bool), stroke_color, stroke_width). closed=False: the same outlines without their
    pathClose (open sub-paths: the strokes get caps; the fills are the same polygons)."""
    rs = np.random.RandomState(seed)
    b = PathSetBuilder()
    ops = []
    for p in range(npaths):
        b.begin_path()
        nsub = int(rs.choice([1, 2, 3], p=[0.7, 0.2, 0.1]))
        radius = float(np.exp(rs.uniform(np.log(4.0), np.log(120.0))))
        cx0, cy0 = rs.uniform(100.0, 800.0, size=2)
        for s in range(nsub):
            m = int(rs.randint(4, 25))  # cubic segments in this closed sub-path
            cx = cx0 + rs.uniform(-radius, radius) * 0.5
            cy = cy0 + rs.uniform(-radius, radius) * 0.5
            r_s = radius * rs.uniform(0.4, 1.0)
            ang = np.sort(rs.uniform(0.0, 2.0 * np.pi, size=m)) + rs.uniform(0, 2 * np.pi)
            rad = r_s * rs.uniform(0.55, 1.0, size=m)
            P = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], axis=1).astype(np.float32).astype(np.float64)
            # closed Catmull-Rom spline -> cubic beziers
            b.move_to(P[0, 0], P[0, 1])
            for i in range(m):
                p0, p1, p2, p3 = P[(i - 1) % m], P[i], P[(i + 1) % m], P[(i + 2) % m]
                c1 = p1 + (p2 - p0) / 6.0
                c2 = p2 - (p3 - p1) / 6.0
                b.cubic_to(c1[0], c1[1], c2[0], c2[1], p2[0], p2[1])
            if closed:
                b.close()
        b.end_path()
        stroke = bool(rs.uniform() < (1.0 / 3.0))
        ops.append(dict(fill_color=TIGER_PALETTE[int(rs.randint(0, 16))], stroke=stroke,
                        stroke_color=TIGER_PALETTE[int(rs.randint(0, 16))],
                        stroke_width=float(rs.choice([0.5, 0.75, 1.0, 1.5, 2.0, 3.0]))))
    return b.arrays(), ops


def tiger_draws(ops, instances, first_instance=0, join=capi.JOIN_MITER, stretch=False):
    """Draw records for `instances` copies of the drawing; instance i is translated by
    (37*(i%100), 41*(i//100)) at scale 1. Draw order = instance-major. join: the strokes' LineJoin.
    stretch: instance i is also stretched by (1 + e, 1 - e), e in {-16 .. 16} / 64 -- what a caller's transformScale(1 + e, 1 - e) leaves in
    the State; avgScale stays exactly 1 (vg.cpp:4927-4935), so flatten tolerance and stroke widths stay the template's. It changes the
    angles between the segments, i.e. what Round joins count their arc points on (the transformed polyline, stroker.cpp:1146, 1592): the
    instances then differ in SIZE."""
    npaths = len(ops)
    one = make_draws(npaths)
    one["path"] = np.arange(npaths, dtype=np.uint32)
    for p, op in enumerate(ops):
        set_fill(one, p, op["fill_color"], aa=True)
        if op["stroke"]:
            set_stroke(one, p, op["stroke_color"], op["stroke_width"], capi.CAP_BUTT, join, aa=True)
    d = np.tile(one, instances)
    inst = np.repeat(np.arange(first_instance, first_instance + instances, dtype=np.int64), npaths)
    d["mtx"][:, 4] = (37.0 * (inst % 100)).astype(np.float32)
    d["mtx"][:, 5] = (41.0 * (inst // 100)).astype(np.float32)
    if stretch:
        e = (((inst * 37) % 33) - 16).astype(np.float32) / np.float32(64.0)
        d["mtx"][:, 0] = np.float32(1.0) + e
        d["mtx"][:, 3] = np.float32(1.0) - e
    return d


def tiger(instances, seed=2024, first_instance=0):
    ps, ops = tiger_paths(seed)
    return ps, tiger_draws(ops, instances, first_instance)


def tiger_varied_draws(ops, instances, first_instance=0, seed=77, join=capi.JOIN_MITER):
    """Tiger instances that do NOT share one subdivision: instance i is drawn at scale s_i in {0.5, 1, 1.5, 2, 2.5, 3, 3.5}
    under a rotation, i.e. what a caller's State would hold after transformScale / transformRotate / transformTranslate
    (updateState: avgScale = mean of the column norms, vg.cpp:4927-4935). The flatten tolerance and the stroke widths
    follow the scale (pathReset(avgScale), ctxStrokePathColor :3416-3433), so the 64 lanes of an instanced wave walk
    different subdivisions. Same draw order as tiger_draws."""
    npaths = len(ops)
    rs = np.random.RandomState(seed)
    scales = np.float32([0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5])
    d = np.zeros(0, dtype=capi.draw_dtype)
    inst = np.arange(first_instance, first_instance + instances, dtype=np.int64)
    sc = scales[rs.randint(0, len(scales), size=first_instance + instances)[first_instance:]]
    ang = rs.uniform(0.0, 2.0 * np.pi, size=first_instance + instances)[first_instance:]
    c, sn = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    m = np.zeros((instances, 6), np.float32)
    m[:, 0] = sc * c; m[:, 1] = sc * sn; m[:, 2] = -(sc * sn); m[:, 3] = sc * c
    m[:, 4] = (37.0 * (inst % 100)).astype(np.float32)
    m[:, 5] = (41.0 * (inst // 100)).astype(np.float32)
    avg = ((np.sqrt(m[:, 0] * m[:, 0] + m[:, 2] * m[:, 2]) + np.sqrt(m[:, 1] * m[:, 1] + m[:, 3] * m[:, 3])) * np.float32(0.5)).astype(np.float32)
    # one table of per-path records per distinct scale value (set_stroke is scalar in the scale)
    tables = {}
    for a in np.unique(avg):
        one = make_draws(npaths)
        one["path"] = np.arange(npaths, dtype=np.uint32)
        one["scale"] = a
        for p, op in enumerate(ops):
            set_fill(one, p, op["fill_color"], aa=True)
            if op["stroke"]:
                set_stroke(one, p, op["stroke_color"], op["stroke_width"], capi.CAP_BUTT, join, aa=True, avg_scale=float(a))
        tables[float(a)] = one
    d = np.concatenate([tables[float(a)] for a in avg])
    d["mtx"] = np.repeat(m, npaths, axis=0)
    return d


def tiger_spec_paths(seed=2025, npaths=240):
    """The drawing of SURVEY.md 8(d) config 3 as written: 240 paths, each 1-4 closed sub-paths of 8-60 cubic segments
    (smooth random closed splines in a 900^2 view box), about 2/3 fill-only and 1/3 fill + stroke (widths 0.5-3: a mix of
    Thin and AA strokes), colours from a 16-entry palette. Heavier than tiger_paths (which keeps the round-1 shape so that
    rounds compare): about 2.6x the commands per instance."""
    rs = np.random.RandomState(seed)
    b = PathSetBuilder()
    ops = []
    for p in range(npaths):
        b.begin_path()
        nsub = int(rs.randint(1, 5))
        radius = float(np.exp(rs.uniform(np.log(6.0), np.log(140.0))))
        cx0, cy0 = rs.uniform(120.0, 780.0, size=2)
        for s in range(nsub):
            m = int(rs.randint(8, 61))
            cx = cx0 + rs.uniform(-radius, radius) * 0.5
            cy = cy0 + rs.uniform(-radius, radius) * 0.5
            r_s = radius * rs.uniform(0.4, 1.0)
            ang = np.sort(rs.uniform(0.0, 2.0 * np.pi, size=m)) + rs.uniform(0, 2 * np.pi)
            rad = r_s * rs.uniform(0.6, 1.0, size=m)
            P = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], axis=1).astype(np.float32).astype(np.float64)
            b.move_to(P[0, 0], P[0, 1])
            for i in range(m):
                p0, p1, p2, p3 = P[(i - 1) % m], P[i], P[(i + 1) % m], P[(i + 2) % m]
                c1 = p1 + (p2 - p0) / 6.0
                c2 = p2 - (p3 - p1) / 6.0
                b.cubic_to(c1[0], c1[1], c2[0], c2[1], p2[0], p2[1])
            b.close()
        b.end_path()
        stroke = bool(rs.uniform() < (1.0 / 3.0))
        ops.append(dict(fill_color=TIGER_PALETTE[int(rs.randint(0, 16))], stroke=stroke,
                        stroke_color=TIGER_PALETTE[int(rs.randint(0, 16))],
                        stroke_width=float(rs.uniform(0.5, 3.0))))
    return b.arrays(), ops


# ---- config 3 ----------------------------------------------------------------------------------
def random_walk_polylines(n=10000, nseg=1000, seed=5678, width=6.0, cap=capi.CAP_ROUND, join=capi.JOIN_ROUND,
                          step=8.0, turn_sigma=0.5):
    """n open polylines of nseg segments: start uniform in [0,1000)^2, fixed step, heading += N(0, sigma)."""
    rs = np.random.RandomState(seed)
    start = rs.uniform(0.0, 1000.0, size=(n, 2))
    heading = np.cumsum(rs.normal(0.0, turn_sigma, size=(n, nseg)), axis=1) + rs.uniform(0, 2 * np.pi, size=(n, 1))
    dxy = np.stack([np.cos(heading), np.sin(heading)], axis=2) * step
    pts = np.concatenate([start[:, None, :], start[:, None, :] + np.cumsum(dxy, axis=1)], axis=1).astype(np.float32)
    ncmd_per = nseg + 1
    cmd_type = np.full((n, ncmd_per), capi.CMD_LINE_TO, dtype=np.uint8)
    cmd_type[:, 0] = capi.CMD_MOVE_TO
    arg_off = np.arange(n * ncmd_per + 1, dtype=np.uint32) * 2
    path_begin = np.arange(n + 1, dtype=np.uint32) * ncmd_per
    from .pathset import PathSetArrays
    ps = PathSetArrays(cmd_type.reshape(-1), arg_off, pts.reshape(-1), path_begin)
    d = make_draws(n)
    d["path"] = np.arange(n, dtype=np.uint32)
    set_stroke(d, slice(None), 0xFF2080FF, width, cap, join, aa=True)
    return ps, d


# ---- mixed fuzz set used by the parity tests ---------------------------------------------------
def fuzz_paths(seed, npaths=64, with_shapes=True, degenerate=True, with_polylines=None):
    """Random paths exercising every command and the degenerate cases the reference has branches for
    (zero-length segments, coincident control points, closing onto the start point, tiny curves)."""
    if with_polylines is None:
        with_polylines = with_shapes
    rs = np.random.RandomState(seed)
    b = PathSetBuilder()
    for p in range(npaths):
        b.begin_path()
        nsub = int(rs.randint(1, 4))
        prev_open = True  # an ARC may only lead a sub-path at the path start or after an OPEN sub-path
        for s in range(nsub):
            kind = int(rs.randint(0, 10)) if with_shapes else 0
            if kind == 9 and not prev_open:
                kind = 0
            if kind in (6, 7, 8):
                prev_open = False
            scale = float(rs.choice([1.0, 10.0, 100.0, 400.0]))
            ox, oy = rs.uniform(-50, 50, size=2)
            if kind == 6:
                b.rect(ox, oy, rs.uniform(-1, 1) * scale, rs.uniform(-1, 1) * scale)
                continue
            if kind == 7:
                w, h = rs.uniform(0.2, 1, size=2) * scale
                if rs.uniform() < 0.3:
                    h = w
                if rs.uniform() < 0.5:
                    b.rounded_rect(ox, oy, w, h, rs.uniform(0, 0.6) * scale)
                else:
                    r4 = rs.uniform(0, 0.6, size=4) * scale * (rs.uniform(size=4) < 0.8)
                    b.rounded_rect_varying(ox, oy, w, h, *r4)
                continue
            if kind == 8:
                if rs.uniform() < 0.5:
                    b.circle(ox, oy, rs.uniform(0.05, 1) * scale)
                else:
                    b.ellipse(ox, oy, rs.uniform(0.05, 1) * scale, rs.uniform(0.05, 1) * scale)
                continue
            if kind == 9:
                b.arc(ox, oy, rs.uniform(0.1, 1) * scale, rs.uniform(-7, 7), rs.uniform(-7, 7), rs.uniform() < 0.5)
                ncont = int(rs.randint(0, 3))
            else:
                b.move_to(ox, oy)
                ncont = int(rs.randint(1, 12))
            cx, cy = ox, oy
            for c in range(ncont):
                t = int(rs.randint(0, 8))
                nx, ny = cx + rs.uniform(-1, 1) * scale, cy + rs.uniform(-1, 1) * scale
                if degenerate and rs.uniform() < 0.08:
                    nx, ny = cx, cy  # zero-length step
                if degenerate and rs.uniform() < 0.05:
                    nx, ny = cx + 1e-3, cy - 2e-3  # inside the epsilon ball
                if t <= 1:
                    b.line_to(nx, ny)
                elif t <= 4:
                    c1 = (cx + rs.uniform(-1, 1) * scale, cy + rs.uniform(-1, 1) * scale)
                    c2 = (nx + rs.uniform(-1, 1) * scale, ny + rs.uniform(-1, 1) * scale)
                    if degenerate and rs.uniform() < 0.1:
                        c1 = (cx, cy)
                    if degenerate and rs.uniform() < 0.1:
                        c2 = (nx, ny)
                    b.cubic_to(c1[0], c1[1], c2[0], c2[1], nx, ny)
                elif t == 5:
                    b.quadratic_to(cx + rs.uniform(-1, 1) * scale, cy + rs.uniform(-1, 1) * scale, nx, ny)
                elif t == 6 and with_shapes:
                    b.arc_to(cx + rs.uniform(-1, 1) * scale, cy + rs.uniform(-1, 1) * scale, nx, ny, rs.uniform(0.05, 0.5) * scale)
                    nx, ny = None, None
                elif t == 7 and with_polylines:
                    k = int(rs.randint(1, 6))
                    pts = np.cumsum(rs.uniform(-1, 1, size=(k, 2)) * scale, axis=0) + np.array([cx, cy])
                    if degenerate and rs.uniform() < 0.3:
                        pts[0] = (cx, cy)
                    b.polyline(pts)
                    nx, ny = float(np.float32(pts[-1, 0])), float(np.float32(pts[-1, 1]))
                else:
                    b.line_to(nx, ny)
                if nx is None:
                    # arcTo end point is computed; continue from a fresh random point
                    cx, cy = cx + rs.uniform(-1, 1) * scale, cy + rs.uniform(-1, 1) * scale
                else:
                    cx, cy = nx, ny
            r = rs.uniform()
            prev_open = True
            if kind == 9 and ncont == 0:
                pass  # a lone arc stays open (CLOSE directly after ARC is fine too, but keep some open)
            elif r < 0.35:
                b.close()
                prev_open = False
            elif r < 0.5 and ncont >= 2 and kind != 9:
                b.line_to(ox, oy)  # return exactly to the start, then close (pathClose pops it)
                b.close()
                prev_open = False
        b.end_path()
    return b.arrays()


def fuzz_draws(ps, seed, ndraws=None):
    """Random op/parameter assignment over every stroker entry point the reference exposes."""
    rs = np.random.RandomState(seed + 77)
    n = ps.npaths if ndraws is None else ndraws
    d = make_draws(n)
    d["path"] = np.arange(n, dtype=np.uint32) % ps.npaths
    for i in range(n):
        sc = float(rs.choice([0.5, 1.0, 1.0, 2.0, 7.5]))
        ang = rs.uniform(0, 2 * np.pi)
        d["mtx"][i] = [sc * np.cos(ang), sc * np.sin(ang), -sc * np.sin(ang), sc * np.cos(ang), rs.uniform(-100, 100), rs.uniform(-100, 100)]
        d["scale"][i] = np.float32(sc)
        d["tess_tol"][i] = np.float32(rs.choice([0.25, 0.25, 0.1, 0.5]))
        d["fringe"][i] = np.float32(rs.choice([1.0, 1.0, 0.5]))
        if rs.uniform() < 0.6:
            set_fill(d, i, int(rs.randint(0, 1 << 32, dtype=np.uint64)), aa=bool(rs.uniform() < 0.75))
        if rs.uniform() < 0.8:
            set_stroke(d, i, int(rs.randint(0, 1 << 32, dtype=np.uint64)), float(rs.choice([0.3, 0.9, 1.5, 3.0, 10.0, 40.0])),
                       int(rs.randint(0, 3)), int(rs.randint(0, 3)), aa=bool(rs.uniform() < 0.75), avg_scale=sc,
                       fringe=float(d["fringe"][i]))
    return d


def thin_fuzz_paths(seed, npaths=64, degenerate=True):
    """Random paths of moveTo / lineTo / close only (what vgx_thin.h lays out statically): one to four sub-paths, open and closed,
    sub-paths of one or two vertices (a lone moveTo, moveTo + close, moveTo + lineTo + close), polygons that return exactly -- or
    to within the epsilon of pathClose -- to their first point before closing (the vertex pathClose pops), long runs (more than
    one 64-command chunk) and, with `degenerate`, lineTo commands onto the current point (pathLineTo drops them: the exact builder
    takes such draws)."""
    rs = np.random.RandomState(seed + 4242)
    b = PathSetBuilder()
    for p in range(npaths):
        b.begin_path()
        for s in range(int(rs.randint(1, 5))):
            scale = float(rs.choice([1.0, 10.0, 100.0, 400.0]))
            ox, oy = [float(np.float32(v)) for v in rs.uniform(-50, 50, size=2)]
            b.move_to(ox, oy)
            r = rs.uniform()
            n = 0 if r < 0.12 else (1 if r < 0.25 else (int(rs.randint(2, 12)) if r < 0.9 else int(rs.randint(60, 300))))
            x, y = ox, oy
            for i in range(n):
                x = float(np.float32(x + rs.uniform(-1, 1) * scale)); y = float(np.float32(y + rs.uniform(-1, 1) * scale))
                b.line_to(x, y)
                if degenerate and rs.uniform() < 0.02:
                    b.line_to(x, y)  # zero-length segment
            c = rs.uniform()
            if c < 0.3:
                b.close()
            elif c < 0.5:
                b.line_to(ox, oy); b.close()  # back onto the first point, then close: popped when the sub-path has more than two vertices
            elif c < 0.6:
                b.line_to(float(np.float32(ox + 1.0e-4)), oy); b.close()  # within pathClose's epsilon of the first point
        b.end_path()
    return b.arrays()


# ---- closed-shape fuzz set (template mode: fills + closed Miter AA strokes) ----------------------
def closed_fuzz_paths(seed, npaths=48):
    """Random paths whose sub-paths are all CLOSED with at least three distinct vertices: move / line / cubic / quad /
    polyline runs ending in CLOSE, rects, rounded rects, circles, ellipses -- i.e. every stroke of them is a closed
    stroke (the kind vgx_tmpl.hip emits) and every command of the flattener is exercised."""
    rs = np.random.RandomState(seed)
    b = PathSetBuilder()
    for p in range(npaths):
        b.begin_path()
        for s in range(int(rs.randint(1, 4))):
            kind = int(rs.randint(0, 8))
            scale = float(rs.choice([3.0, 20.0, 120.0]))
            ox, oy = rs.uniform(-80, 80, size=2)
            if kind == 5:
                b.rect(ox, oy, rs.uniform(0.3, 1) * scale, rs.uniform(0.3, 1) * scale)
            elif kind == 6:
                w, h = rs.uniform(0.4, 1, size=2) * scale
                if rs.uniform() < 0.5:
                    b.rounded_rect(ox, oy, w, h, rs.uniform(0.05, 0.4) * scale)
                else:
                    b.rounded_rect_varying(ox, oy, w, h, *(rs.uniform(0.02, 0.4, size=4) * scale))
            elif kind == 7:
                if rs.uniform() < 0.5:
                    b.circle(ox, oy, rs.uniform(0.2, 1) * scale)
                else:
                    b.ellipse(ox, oy, rs.uniform(0.2, 1) * scale, rs.uniform(0.2, 1) * scale)
            else:
                m = int(rs.randint(3, 14))
                ang = np.sort(rs.uniform(0.0, 2.0 * np.pi, size=m))
                rad = scale * rs.uniform(0.5, 1.0, size=m)
                P = np.stack([ox + rad * np.cos(ang), oy + rad * np.sin(ang)], axis=1)
                b.move_to(P[0, 0], P[0, 1])
                for i in range(1, m + 1):
                    q = P[i % m]
                    a = P[i - 1]
                    t = int(rs.randint(0, 5))
                    if i == m and rs.uniform() < 0.5:
                        break  # leave the last edge to CLOSE
                    if t <= 1:
                        b.line_to(q[0], q[1])
                    elif t <= 3:
                        c1 = a + (q - a) * 0.33 + rs.uniform(-0.2, 0.2, size=2) * scale
                        c2 = a + (q - a) * 0.66 + rs.uniform(-0.2, 0.2, size=2) * scale
                        b.cubic_to(c1[0], c1[1], c2[0], c2[1], q[0], q[1])
                    else:
                        c = (a + q) * 0.5 + rs.uniform(-0.3, 0.3, size=2) * scale
                        b.quadratic_to(c[0], c[1], q[0], q[1])
                b.close()
        b.end_path()
    return b.arrays()


def template_draws(ps, seed, instances, same_colors=False):
    """One drawing of `ps` (every path once; fills AA / plain / none, closed Miter AA strokes of hairline and regular
    widths, per-path scale / tolerance / fringe) repeated for `instances` instances that differ only in what the template
    mode allows to differ: the transform (any affine matrix; the record's `scale` stays the path's) and the colours."""
    rs = np.random.RandomState(seed + 4242)
    n = ps.npaths
    one = make_draws(n)
    one["path"] = np.arange(n, dtype=np.uint32)
    for i in range(n):
        sc = float(rs.choice([0.5, 1.0, 1.0, 2.0]))
        one["scale"][i] = np.float32(sc)
        one["tess_tol"][i] = np.float32(rs.choice([0.25, 0.25, 0.1, 0.5]))
        one["fringe"][i] = np.float32(rs.choice([1.0, 1.0, 0.5]))
        r = rs.uniform()
        if r < 0.8:
            set_fill(one, i, int(rs.randint(0, 1 << 32, dtype=np.uint64)), aa=bool(rs.uniform() < 0.8))
            if rs.uniform() < 0.2:
                one["fill_flags"][i] |= np.uint32(0x100)  # VGX_FILL_INDEX_ORDER_SSE
        if r > 0.5 or rs.uniform() < 0.3:
            set_stroke(one, i, int(rs.randint(0, 1 << 32, dtype=np.uint64)), float(rs.choice([0.3, 0.9, 1.5, 3.0, 12.0])),
                       capi.CAP_BUTT, capi.JOIN_MITER, aa=True, avg_scale=sc, fringe=float(one["fringe"][i]))
    d = np.tile(one, instances)
    for k in range(instances):
        s = slice(k * n, (k + 1) * n)
        f = float(rs.choice([0.5, 1.0, 1.0, 3.0]))
        ang = rs.uniform(0, 2 * np.pi)
        sh = rs.uniform(-0.3, 0.3)
        c, sn = np.cos(ang), np.sin(ang)
        d["mtx"][s] = np.float32([f * c, f * sn, f * (-sn + sh * c), f * (c + sh * sn), rs.uniform(-500, 500), rs.uniform(-500, 500)])
        if k % 7 == 3:
            d["mtx"][s, 0] *= np.float32(-1.0)  # mirrored instance: fill orientation and the joins' inner sides flip
            d["mtx"][s, 1] *= np.float32(-1.0)
        if not same_colors:
            d["fill_color"][s] = rs.randint(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
            d["stroke_color"][s] = rs.randint(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    return d


def template_class_draws(ps, seed, instances, nclasses):
    """template_draws in `nclasses` flavours: flavour c is the same drawing with its own per-path scale / tolerance / fringe,
    fill kinds and stroke widths (as if recorded under another State); every instance takes one flavour at random (the first
    `nclasses` instances one each, so that all occur) and its own transform and colours. What the template mode handles with
    one template per class."""
    rs = np.random.RandomState(seed + 9191)
    n = ps.npaths
    flav = [template_draws(ps, seed + 31 * c, 1, same_colors=True) for c in range(nclasses)]
    pick = np.concatenate([np.arange(nclasses), rs.randint(0, nclasses, size=max(0, instances - nclasses))])[:instances]
    base = template_draws(ps, seed, instances)  # transforms / colours per instance
    d = np.concatenate([flav[int(c)] for c in pick])
    d["mtx"] = base["mtx"]
    d["fill_color"] = base["fill_color"]
    d["stroke_color"] = base["stroke_color"]
    return d, pick


def template_class_round_draws(ps, seed, instances, nclasses, closed_aa_only=False):
    """template_class_draws whose flavours hold Round joins (template_general_draws(round_joins=True) per flavour: other scales, widths,
    tolerances AND other stroke styles from class to class): mesh sizes belong to the instance, tables to its class. closed_aa_only: AA
    strokes with Miter / Bevel / Round joins only (with closed paths: the closed-stroke kernels)."""
    rs = np.random.RandomState(seed + 4242)
    flav = []
    for c in range(nclasses):
        if closed_aa_only:
            one = template_draws(ps, seed + 31 * c, 1, same_colors=True)
            r2 = np.random.RandomState(seed + 17 * c)
            for i in range(ps.npaths):
                if r2.uniform() < 0.85:
                    set_stroke(one, i, 0xFF102030, float(r2.choice([1.5, 3.0, 12.0])), capi.CAP_BUTT, int(r2.choice([capi.JOIN_MITER, capi.JOIN_BEVEL, capi.JOIN_ROUND])), aa=True,
                               avg_scale=float(one["scale"][i]), fringe=float(one["fringe"][i]))
            flav.append(one)
        else:
            flav.append(template_general_draws(ps, seed + 31 * c, 1, round_joins=True))
    pick = np.concatenate([np.arange(nclasses), rs.randint(0, nclasses, size=max(0, instances - nclasses))])[:instances]
    base = template_draws(ps, seed, instances)
    d = np.concatenate([flav[int(c)] for c in pick])
    d["mtx"] = base["mtx"]
    d["fill_color"] = base["fill_color"]
    d["stroke_color"] = base["stroke_color"]
    return d, pick


def template_general_draws(ps, seed, instances, round_joins=False):
    """template_draws with every stroke style whose mesh sizes do not depend on the geometry: open and closed sub-paths (whatever
    `ps` holds), Butt / Square / Round caps, Miter / Bevel joins, AA / non-AA / hairline (Thin) strokes."""
    rs = np.random.RandomState(seed + 777)
    d = template_draws(ps, seed, instances)
    n = ps.npaths
    one = d[:n].copy()
    for i in range(n):
        if rs.uniform() < 0.85:
            sc = float(one["scale"][i])
            join = int(rs.choice([capi.JOIN_MITER, capi.JOIN_BEVEL, capi.JOIN_ROUND])) if round_joins else int(rs.choice([capi.JOIN_MITER, capi.JOIN_BEVEL]))
            set_stroke(one, i, int(rs.randint(0, 1 << 32, dtype=np.uint64)), float(rs.choice([0.3, 0.9, 1.5, 3.0, 12.0])),
                       int(rs.choice([capi.CAP_BUTT, capi.CAP_SQUARE, capi.CAP_ROUND])), join, aa=bool(rs.uniform() < 0.75), avg_scale=sc, fringe=float(one["fringe"][i]))
    for k in ("stroke_flags", "stroke_width"):
        d[k] = np.tile(one[k], instances)
    return d

