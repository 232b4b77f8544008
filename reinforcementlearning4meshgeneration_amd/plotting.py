"""PNG of a generated mesh: MeshGeneration.save_meshes (general/mesh.py:1785-1792), the call the evaluation callback makes
after every evaluation episode (rl/baselines/CustomizeCallback.py:131-133) and testbed.py:189-212 makes per domain.

The reference draws ``self.boundary`` -- every segment of the vertex graph, i.e. the domain's ring plus the edges of all
elements generated so far (Boundary2D.plot, general/components.py:255-264) -- and labels the elements passed in
``meshes`` at a shifted centroid (generate_meshes_canvas, general/mesh.py:1762-1783).  Here the graph is rebuilt from the
element log in the order the reference's ``all_segments`` walks it (general/components.py:222-234), so the sequence of
matplotlib calls -- and with it the image -- is the reference's (tests/test_gpu_plotting.py compares with PNGs the
reference wrote).  Host-side drawing only; the numbers on the canvas (get_quality) come from the device.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence

import numpy as np


def mesh_segments(n0: int, quads: np.ndarray, n_vert: Optional[int] = None) -> list:
    """[(a, b)] vertex-id pairs in Boundary2D.all_segments() order for a ring of n0 vertices plus the logged elements.

    Vertex.segments order: the ring is connected as connect_vertices does (general/mesh.py:1926-1930 -- Segment(v[i-1],
    v[i]) assigned to both ends, i ascending), each element as Mesh.connect_vertices does (general/components.py:832-837
    -- Segment(v[i], v[i-1]) unless the two already share one)."""
    nv = max(n0, int(quads.max()) + 1 if quads.size else 0, n_vert or 0)
    segs_of = [[] for _ in range(nv)]
    segs = []

    def connect(a, b):
        segs.append((a, b))
        segs_of[a].append(len(segs) - 1)
        segs_of[b].append(len(segs) - 1)

    for i in range(n0):
        connect((i - 1) % n0, i)
    linked = {frozenset(s) for s in segs}
    for q in quads:
        for i in range(4):
            a, b = int(q[i]), int(q[i - 1])
            if frozenset((a, b)) not in linked:
                linked.add(frozenset((a, b)))
                connect(a, b)
    seen, order = set(), []
    for v in range(nv):
        for s in segs_of[v]:
            if s not in seen:
                seen.add(s)
                order.append(segs[s])
    return order


def label_position(quad_xy) -> tuple:
    """Mesh.get_centriod(diff=True), general/components.py:820-830."""
    q = [(float(p[0]), float(p[1])) for p in quad_xy]
    sx = sy = 0.0
    for x, y in q:
        sx += x
        sy += y
    diff = sum(math.sqrt((q[i][0] - q[i - 1][0]) ** 2 + (q[i][1] - q[i - 1][1]) ** 2) for i in range(4)) / 4
    return sx / 4 - diff / 3, sy / 4 - diff * 0.1


def element_xy(element) -> np.ndarray:
    """[4, 2] float64 of an element given as an array or as a reference-style object with .vertices[k].x / .y."""
    if hasattr(element, "vertices"):
        return np.array([[float(v.x), float(v.y)] for v in element.vertices], np.float64).reshape(4, 2)
    return np.asarray(element, np.float64).reshape(4, 2)


def save_meshes(name, n0: int, quads: np.ndarray, vertex_xy: np.ndarray, meshes: Sequence, quality: bool = False,
                indexing: bool = False, type: int = 0, dpi: int = 300, style: str = "k.-",
                quality_of: Optional[Callable[[Sequence, int], Sequence[float]]] = None) -> None:
    """general/mesh.py:1785-1792.  n0 / quads / vertex_xy: the episode (ring length, element log, vertex table);
    meshes: the elements to label; quality_of(meshes, index) -> get_quality(element, index) of each."""
    import matplotlib.pyplot as plt

    plt.clf()
    # Boundary2D.plot(style=style, linewidth=1), general/components.py:255-264
    fig = plt.figure()
    ax = fig.add_subplot(111)
    xs, ys = [], []
    for a, b in mesh_segments(n0, np.asarray(quads, np.int64).reshape(-1, 4), len(vertex_xy)):
        x1, y1, x2, y2 = (float(vertex_xy[a][0]), float(vertex_xy[a][1]), float(vertex_xy[b][0]), float(vertex_xy[b][1]))
        plt.plot([x1, x2], [y1, y2], style, linewidth=1, markersize=10)   # Segment.show, general/components.py:556-564
        plt.gca().set_aspect("equal", adjustable="box")
        xs.extend([x1, x2])
        ys.extend([y1, y2])
    ax.set_frame_on(False)
    plt.gca().set_xlim([min(xs) - 0.1, max(xs) + 0.1])
    plt.gca().set_ylim([min(ys) - 0.1, max(ys) + 0.1])
    plt.xticks([])
    plt.yticks([])
    # generate_meshes_canvas, general/mesh.py:1762-1783
    meshes = [element_xy(m) for m in meshes]
    values = None
    if quality and len(meshes):
        if quality_of is None:
            raise ValueError("save_meshes(quality=True) needs the element qualities")
        values = [round(float(v), 4) for v in quality_of(meshes, type)]
    for k, m in enumerate(meshes):
        cx, cy = label_position(m)
        if quality and indexing:
            plt.text(cx, cy, f"{k}; {values[k]}", fontsize=6)
        elif quality:
            plt.text(cx, cy, values[k], fontsize=6)
        elif indexing:
            plt.text(cx, cy, k, fontsize=4)
    plt.gca().set_aspect("equal", adjustable="box")
    plt.subplots_adjust(top=1, bottom=0, right=1, left=-0, hspace=0, wspace=0)
    plt.savefig(name, dpi=dpi)
    plt.close("all")
