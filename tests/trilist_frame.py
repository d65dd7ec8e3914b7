"""Frames that contain vg::indexedTriList user meshes (reference src/vg.cpp:4129-4175, command-list form :2566-2611, :4461-4477)
(TEST INFRASTRUCTURE): scripts and the host-side reference composition shared by the CPU and the GPU test."""
import numpy as np

import cmdlist_util as cu
import pyvgref as R
from vgscript import Script

FILL_AA = cu.fill_flags(aa=True)
FILL_CONCAVE_AA = cu.fill_flags(concave=True, aa=True)


def quad(x, y, w, h):
    return np.asarray([[x, y], [x + w, y], [x + w, y + h], [x, y + h]], np.float32), np.asarray([0, 1, 2, 0, 2, 3], np.uint16)


def grid(nx, ny, x0, y0, step, rng):
    """(nx+1)(ny+1) vertices, 6 nx ny indices, per-vertex colours, jittered positions."""
    xs, ys = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1))
    pos = np.stack([x0 + xs.reshape(-1) * step, y0 + ys.reshape(-1) * step], 1).astype(np.float32)
    pos += rng.uniform(-0.3, 0.3, pos.shape).astype(np.float32)
    idx = []
    for j in range(ny):
        for i in range(nx):
            a = j * (nx + 1) + i
            idx += [a, a + 1, a + nx + 2, a, a + nx + 2, a + nx + 1]
    col = rng.integers(0, 2 ** 32, pos.shape[0], dtype=np.uint64).astype(np.uint32)
    return pos, np.asarray(idx, np.uint16), col


def uv_of(pos, uv_float, rng):
    if uv_float:
        return rng.uniform(0, 1, pos.shape).astype(np.float32)
    return rng.integers(-32768, 32767, pos.shape, dtype=np.int64).astype(np.int16)


def s_trilist(uv_float=False, image=3, seed=5):
    """Colour fills around user meshes: one that joins its neighbours' draw command (no image: the font atlas, white-pixel UVs),
    ones with their own image / UVs / per-vertex colours, under transforms, scissors, a saved state, inside an open path."""
    rng = np.random.default_rng(seed)
    s = Script()
    s.begin_path().rect(10, 10, 100, 60).fill(0xFF2040F0, FILL_AA)
    p, i = quad(200, 20, 50, 40)
    s.indexed_tri_list(p, [0xFFFFFFFF], i)                                   # one colour, no UV, no image
    s.begin_path().circle(300, 200, 40).stroke(0xFF00FF00, 3.0, cu.stroke_flags(0, 0))
    s.push().translate(40, 300).rotate(0.3).scale(1.5, 0.75)
    gp, gi, gc = grid(5, 3, 0, 0, 12.0, rng)
    s.indexed_tri_list(gp, gc, gi, uv=uv_of(gp, uv_float, rng), image=image)  # per-vertex colours + UVs + an image
    s.indexed_tri_list(gp + 7, gc[:1], gi[:9], image=image)                   # same image: the same draw command
    s.set_scissor(0, 0, 600, 500)
    s.indexed_tri_list(gp, gc, gi, uv=uv_of(gp, uv_float, rng))               # UVs on the font atlas, another scissor
    s.pop()
    s.begin_path().move_to(400, 50).line_to(500, 60)
    p, i = quad(420, 300, 30, 30)
    s.indexed_tri_list(p, [0x80FF0000], i)                                    # while a path is being built
    s.line_to(480, 150).close_path().fill(0xFF808080, FILL_AA)
    s.indexed_tri_list(np.zeros((0, 2), np.float32), [0xFFFFFFFF], np.zeros(0, np.uint16))  # empty
    s.indexed_tri_list(p[:3] + 100, [1, 2, 3], np.zeros(0, np.uint16))       # vertices without indices
    s.begin_path().move_to(600, 100).line_to(700, 100).line_to(620, 180).line_to(660, 60).line_to(700, 180).close_path().fill(0xFF00FFFF, FILL_CONCAVE_AA)
    p, i = quad(640, 300, 20, 50)
    s.indexed_tri_list(p, [0xFF123456], i, image=image + 1)
    return s


def s_trilist_only(uv_float=False):
    """Nothing but user meshes: no path in the whole list."""
    rng = np.random.default_rng(11)
    s = Script()
    for k in range(4):
        gp, gi, gc = grid(3 + k, 2, 20.0 * k, 30.0, 9.0, rng)
        s.translate(3.0, 2.0)
        s.indexed_tri_list(gp, gc, gi, uv=uv_of(gp, uv_float, rng) if k & 1 else None, image=0xFFFF if k < 2 else 2)
    return s


def s_random(seed, uv_float=False):
    rng = np.random.default_rng(seed)
    s = Script()
    for k in range(int(rng.integers(8, 20))):
        r = rng.random()
        if r < 0.45:
            nx, ny = int(rng.integers(1, 12)), int(rng.integers(1, 12))
            gp, gi, gc = grid(nx, ny, float(rng.uniform(0, 800)), float(rng.uniform(0, 500)), float(rng.uniform(2, 20)), rng)
            one = rng.random() < 0.4
            s.indexed_tri_list(gp, gc[:1] if one else gc, gi, uv=uv_of(gp, uv_float, rng) if rng.random() < 0.5 else None,
                               image=int(rng.choice([0xFFFF, 0xFFFF, 1, 2, 3])))
        elif r < 0.7:
            s.begin_path().rounded_rect(float(rng.uniform(0, 800)), float(rng.uniform(0, 500)), float(rng.uniform(5, 90)), float(rng.uniform(5, 90)), 4.0)
            s.fill(int(rng.integers(0, 2 ** 32)) | 0xFF000000, FILL_AA)
        elif r < 0.85:
            s.begin_path().circle(float(rng.uniform(0, 800)), float(rng.uniform(0, 500)), float(rng.uniform(3, 60)))
            s.stroke(int(rng.integers(0, 2 ** 32)) | 0xFF000000, float(rng.uniform(0.5, 6)), cu.stroke_flags(int(rng.integers(0, 3)), int(rng.integers(0, 3))))
        elif r < 0.93:
            s.translate(float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))).rotate(float(rng.uniform(-0.2, 0.2)))
        else:
            s.set_scissor(float(rng.uniform(0, 100)), float(rng.uniform(0, 100)), float(rng.uniform(300, 900)), float(rng.uniform(300, 600)))
    return s


def make_images(rc, n=6):
    """User images so that image handles 1.. exist in the reference Context (0 is the font atlas)."""
    return [rc.create_image(8, 8) for _ in range(n)]
