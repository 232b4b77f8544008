"""Domain polygons and the per-domain constants the reference computes in __init__/reset.

Host-side (Python) part of the hot path's input format:

* ``boundary(index)``        -- the built-in domains of general/polygon.py:76-108 (literal vertex data)
* ``read_polygon(path)``     -- ui/domains/*.json loader, general/polygon.py:110-117 (first line of the
                               file is ``[[x, y], ...]`` in pixels; coordinates are divided by 100)
* ``generate_polygon(...)``  -- restatement of ui/GenerateRandomPolygon.py:5-49 (config 5)
* ``calculate_density`` / ``check_clockwise`` / ``normalise_clockwise`` / ``density_domain`` -- the drawing UI's graded
                               edge subdivision and orientation rule (ui/tk-ui.py:252-276, 169-176, 84-101)
* ``domain_constants(pts)``  -- original_area (general/components.py:477-479), average_edge_length
                               (components.py:442-447) and estimate_area_range (general/mesh.py:679-692)

Everything ``BoudaryEnv.reset()`` computes besides these three is recomputed on the device by the
reset kernel.  The constants are computed here with the same Python/NumPy expressions the reference
uses (``**`` is libm pow, ``sum`` is a left-to-right loop, ``np.dot`` is BLAS), so that they are the
reference's numbers and not a re-derivation.
"""
from __future__ import annotations

import json
import math
import random
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

Point = Tuple[float, float]

# general/polygon.py:79-83 -- data
_BOUNDARY_0 = [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6),
               (1, 6), (2, 6), (3, 6), (4, 6), (5, 6), (6, 6),
               (7, 5), (8, 4), (9, 3), (10, 2), (11, 1), (12, 0),
               (11, -1), (10, -2), (9, -3), (8, -4), (7, -5), (6, -6),
               (5, -5), (4, -4), (3, -3), (2, -2), (1, -1), (0, 0)]
# general/polygon.py:84-93
_BOUNDARY_1 = [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6), (0, 7), (0, 8), (0, 9), (0, 10), (0, 11),
               (1, 11), (2, 11), (3, 11), (4, 11), (5, 11), (6, 11), (7, 11), (8, 11),
               (9, 11), (8, 10), (7.5, 9), (7, 8), (6.5, 7), (6, 6), (6.5, 5.5),
               (7, 5), (8, 4), (9, 3), (10, 2), (11, 1), (12, 0),
               (11, -1), (10, -2), (9, -3), (8, -4), (7, -5), (6, -6),
               (5, -5), (4, -4), (3, -3), (2, -2), (1, -1), (0, 0)]
# general/polygon.py:94-100
_BOUNDARY_2 = [(0, 1), (0, 2), (0, 3), (0, 4), (1, 4), (2, 4), (3, 4), (4, 4),
               (5, 4), (5, 3), (5, 2), (5, 1), (5, 0), (5, -1), (5, -2), (5, -3),
               (5, -4), (5, -5), (5, -6), (5, -7), (4, -7), (3, -7), (2, -7), (1, -7),
               (0, -7), (0, -6), (0, -5), (0, -4), (1, -4), (2, -4), (3, -4), (4, -4),
               (4, -3), (4, -2), (4, -1), (4, 0), (3, 0), (2, 0), (1, 0), (0, 0)]
# general/polygon.py:101-105
_BOUNDARY_M1 = [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6),
                (1, 6), (2, 6), (3, 6), (4, 6), (5, 6), (6, 6),
                (6, 5), (6, 4), (6, 3), (6, 2), (6, 1), (6, 0),
                (5, 0), (4, 0), (3, 0), (2, 0), (1, 0), (0, 0)]


def boundary(index: int = 0) -> List[Point]:
    """Vertex list of general/polygon.py::boundary(index) (clockwise)."""
    table = {0: _BOUNDARY_0, 1: _BOUNDARY_1, 2: _BOUNDARY_2, -1: _BOUNDARY_M1}
    if index not in table:
        raise ValueError(f"boundary(): unknown index {index}")
    return list(table[index])


def read_polygon(filename) -> List[Point]:
    """general/polygon.py:110-117: first line of the file is the JSON vertex list; coords / 100."""
    with open(filename, "r") as fr:
        vertices = json.loads(fr.readline())
    return [(p[0] / 100, p[1] / 100) for p in vertices]


def signed_area2(points: Sequence[Point]) -> float:
    """Twice the shoelace area; negative for the clockwise rings the reference works on."""
    s = 0.0
    n = len(points)
    for i in range(n):
        x0, y0 = points[i - 1]
        x1, y1 = points[i]
        s += x0 * y1 - x1 * y0
    return s


def _clip(x, lo, hi):
    if lo > hi:
        return x
    return lo if x < lo else hi if x > hi else x


def generate_polygon(ctr_x=250, ctr_y=250, ave_radius=100, irregularity=0.55, spikeyness=0.7,
                     num_verts=16, rng: random.Random | None = None) -> List[Tuple[int, int]]:
    """ui/GenerateRandomPolygon.py:5-49 with an explicit ``random.Random`` (pixel coords, CCW)."""
    rng = rng or random
    irregularity = _clip(irregularity, 0, 1) * 2 * math.pi / num_verts
    spikeyness = _clip(spikeyness, 0, 1) * ave_radius
    lower = (2 * math.pi / num_verts) - irregularity
    upper = (2 * math.pi / num_verts) + irregularity
    steps, total = [], 0
    for _ in range(num_verts):
        tmp = rng.uniform(lower, upper)
        steps.append(tmp)
        total = total + tmp
    k = total / (2 * math.pi)
    steps = [s / k for s in steps]
    points = []
    angle = rng.uniform(0, 2 * math.pi)
    for i in range(num_verts):
        r_i = _clip(rng.gauss(ave_radius, spikeyness), 0, 2 * ave_radius)
        points.append((int(ctr_x + r_i * math.cos(angle)), int(ctr_y + r_i * math.sin(angle))))
        angle = angle + steps[i]
    return points


def clockwise_angle(a: Point, b: Point) -> float:
    """ui/tk-ui.py:185-192: direction of a -> b as an angle in [0, 2 pi) (screen coordinates: -atan2(-dy, dx))."""
    theta = -math.atan2(-(b[1] - a[1]), b[0] - a[0])
    return theta if math.copysign(1, theta) >= 0 else 2 * math.pi + theta


def calculate_density(points: Sequence[Point], base_length: float, densities: Sequence[float]) -> List[Point]:
    """The drawing UI's graded edge subdivision, ui/tk-ui.py:252-276 (Density.calculate_density), as a pure function.

    Every vertex carries a density; on the edge prev -> k the spacing grows from A = density(prev) * base_length to
    B = density(k) * base_length in an arithmetic progression: x = round((2L - A - B) / (A + B)) interior points
    (Python's round: half to even), increment e = (B - A) / x, cumulative offsets A (j + 1) + e (j*j + j) / 2, then the
    edge length L itself; the points are prev + offset * (cos, sin)(clockwise_angle(prev, k)).  The first edge is
    (last vertex -> vertex 0).  On the last edge one middle point is dropped if that makes the total EVEN (a quad front
    closes on a 4-ring only with an even vertex count).  Like the reference this raises ZeroDivisionError when an edge
    gets x == 0 (its length within [0.5, 1.5] mean spacings), and vertices are keyed by coordinates (a repeated point
    keeps its first position and its last density).  Pinned by tests/golden/domain_pipeline.json."""
    base_length = float(base_length)
    table = {}
    for i, p in enumerate(points):
        table[tuple(p)] = float(densities[i])
    lst = list(table.items())
    res: List[Point] = []
    for i, (k, v) in enumerate(lst):
        prev, pv = lst[i - 1]
        B = v * base_length
        A = pv * base_length
        L = math.sqrt((prev[0] - k[0]) ** 2 + (prev[1] - k[1]) ** 2)
        x = round((2 * L - A - B) / (A + B))
        e = (B - A) / x
        angle = clockwise_angle(prev, k)
        inter = [A * (j + 1) + e * (j ** 2 + j) / 2 for j in range(x)]
        inter.append(L)
        if i == len(lst) - 1 and (len(res) + len(inter)) % 2 == 1:
            inter.pop(int(len(inter) / 2))
        res.extend((prev[0] + t * math.cos(angle), prev[1] + t * math.sin(angle)) for t in inter)
    return res


def check_clockwise(points: Sequence[Point]) -> bool:
    """ui/tk-ui.py:169-176: the shoelace sum  sum(x[i-1] * y[i] - y[i-1] * x[i])  is negative."""
    return sum([points[i - 1][0] * p[1] - points[i - 1][1] * p[0] for i, p in enumerate(points)]) < 0


def normalise_clockwise(points: Sequence[Point]) -> List[Point]:
    """What file_save writes (ui/tk-ui.py:84-101): the point list as is when check_clockwise(), reversed otherwise."""
    pts = list(points)
    return pts if check_clockwise(pts) else list(reversed(pts))


def density_domain(points: Sequence[Point], base_length: float, densities: Sequence[float] | None = None) -> List[Point]:
    """The reference's whole route from a drawn / generated pixel polygon to a domain ring: calculate_density ->
    save with clockwise normalisation -> read_polygon's division by 100 (general/polygon.py:110-117)."""
    dens = [1.0] * len(points) if densities is None else densities
    return [(p[0] / 100, p[1] / 100) for p in normalise_clockwise(calculate_density(points, base_length, dens))]


def densify(points: Sequence[Point], target: float, even: bool = False) -> List[Point]:
    """Uniform edge split: every edge into ceil(length / target) equal pieces.  NOT the reference's
    calculate_density (that is the function above, which raises on edges of 0.5-1.5 spacings and therefore cannot be
    applied to arbitrary generated polygons): this is the total, always-defined densification bench.py's synthetic
    config-5 rings use.  even=True gives the last edge one more piece when the vertex count would be odd."""
    n = len(points)
    pieces = []
    for i in range(n):
        x0, y0 = points[i]
        x1, y1 = points[(i + 1) % n]
        dx, dy = x1 - x0, y1 - y0
        length = math.sqrt(dx * dx + dy * dy)
        pieces.append(max(1, int(math.ceil(length / target))))
    if even and sum(pieces) % 2 == 1:
        pieces[-1] += 1
    out: List[Point] = []
    for i in range(n):
        x0, y0 = points[i]
        x1, y1 = points[(i + 1) % n]
        for j in range(pieces[i]):
            t = j / pieces[i]
            out.append((x0 + (x1 - x0) * t, y0 + (y1 - y0) * t))
    return out


def random_polygon_px(seed: int, num_verts: int | None = None) -> List[Tuple[int, int]]:
    """The raw polygon of a config-5 domain: generate_polygon (pixel coordinates, CCW) from random.Random(seed) with
    num_verts = randint(8, 64) unless given, consecutive duplicates (int() truncation) dropped; regenerated from the
    same stream until at least 5 distinct vertices remain."""
    rng = random.Random(seed)
    nv = num_verts if num_verts is not None else rng.randint(8, 64)
    while True:
        raw = generate_polygon(num_verts=nv, rng=rng)
        pts: List[Tuple[int, int]] = []
        for p in raw:
            if not pts or p != pts[-1]:
                pts.append(p)
        if len(pts) > 1 and pts[0] == pts[-1]:
            pts.pop()
        if len(pts) >= 5:
            return pts


def random_domain(seed: int, num_verts: int | None = None, edge: float = 0.45) -> List[Point]:
    """Config-5 style synthetic domain: random_polygon_px / 100, made clockwise with the reference's orientation rule
    (normalise_clockwise), uniformly densified to ``edge`` with an even vertex count, coordinates on the 1e-4 grid
    (like every vertex the environment itself creates).  The generator is the reference's (pinned by fixture); the
    densification is the uniform split above, not calculate_density -- see densify()."""
    pts = [(p[0] / 100, p[1] / 100) for p in random_polygon_px(seed, num_verts)]
    pts = densify(normalise_clockwise(pts), edge, even=True)
    return [(round(x, 4), round(y, 4)) for x, y in pts]


def random_density_domain(seed: int, base_length: float = 45.0, density: float = 1.0,
                          num_verts: int | None = None) -> List[Point]:
    """Config-5 domain by the REFERENCE's own route: random_polygon_px(seed) -> calculate_density (every vertex at
    ``density``, ``base_length`` in pixels) -> clockwise rule -> / 100.  Raises ZeroDivisionError where the reference's
    calculate_density does (an edge of 0.5 - 1.5 spacings).  meshenv_create_random_density computes the same ring on the
    device, bit for bit (tests/test_gpu_domgen.py)."""
    px = random_polygon_px(seed, num_verts)
    return density_domain(px, base_length, [density] * len(px))


def device_density_rings(polygons_px: Sequence[Sequence[Tuple[int, int]]], base_length: float,
                         densities: Sequence[Sequence[float]] | None = None, device: int = 0):
    """calculate_density + clockwise rule + / 100 on the DEVICE for explicit pixel polygons (meshenv_density_rings).
    Returns (rings, status): rings[k] = [n_k, 2] float64 array, or None where status[k] != 0 (1: the reference raises
    ZeroDivisionError; 2: more than 2048 ring points; 3: fewer than 3 distinct pixels)."""
    import ctypes as C

    from . import _capi
    L = _capi.load()
    n = len(polygons_px)
    offs = np.zeros(n + 1, np.int32)
    for k, poly in enumerate(polygons_px):
        offs[k + 1] = offs[k] + len(poly)
    flat = [c for poly in polygons_px for p in poly for c in p]
    is_int = 1 if all(isinstance(c, (int, np.integer)) for c in flat) else 0   # Python ints and floats take different arithmetic
    pix = np.ascontiguousarray(flat, np.float64)
    dens = None if densities is None else np.ascontiguousarray([d for ds in densities for d in ds], np.float64)
    cnt = np.zeros(n, np.int32)
    st = np.zeros(n, np.uint8)
    rc = L.meshenv_density_rings(device, n, offs.ctypes.data, pix.ctypes.data, is_int, None if dens is None else dens.ctypes.data,
                                 float(base_length), cnt.ctypes.data, st.ctypes.data, None, C.c_int64(0))
    if rc != 0:
        raise _capi.MeshEnvError(f"meshenv_density_rings failed (code {rc})")
    total = int(cnt[st == 0].sum())
    xy = np.zeros(2 * max(total, 1), np.float64)
    rc = L.meshenv_density_rings(device, n, offs.ctypes.data, pix.ctypes.data, is_int, None if dens is None else dens.ctypes.data,
                                 float(base_length), cnt.ctypes.data, st.ctypes.data, xy.ctypes.data, C.c_int64(total))
    if rc != 0:
        raise _capi.MeshEnvError(f"meshenv_density_rings failed (code {rc})")
    rings, o = [], 0
    for k in range(n):
        if st[k] == 0:
            rings.append(xy[2 * o:2 * (o + int(cnt[k]))].reshape(-1, 2).copy())
            o += int(cnt[k])
        else:
            rings.append(None)
    return rings, st


# --------------------------------------------------------------------------- constants

def _distance(a: Point, b: Point) -> float:
    # Point2D.distance_to, general/components.py:17-18 (`**` kept: it is libm pow, not a*a)
    return math.sqrt((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2)


@dataclass(frozen=True)
class DomainConstants:
    n0: int
    original_area: float
    average_edge_length: float
    est_min_l: float
    est_crit_l: float


def domain_constants(points: Sequence[Point]) -> DomainConstants:
    n = len(points)
    # Boundary2D.poly_area, general/components.py:477-479
    xy = np.array([[x, y] for x, y in points])
    area = 0.5 * np.abs(np.dot(xy[:, 0], np.roll(xy[:, 1], 1)) - np.dot(xy[:, 1], np.roll(xy[:, 0], 1)))
    # Boundary2D.average_edge_length, general/components.py:442-447
    dist = 0
    for i in range(n):
        dist += _distance(points[i], points[i - 1])
    avg = round(dist / n, 4) if n != 0 else 0
    # MeshGeneration.estimate_area_range, general/mesh.py:679-692, over the n ring edges
    lengths = sorted(_distance(points[i - 1], points[i]) for i in range(n))
    total = 0
    for v in lengths:
        total += v
    L = total / len(lengths)
    max_l = min(lengths[-2], 2 * L)
    min_l = min(L / math.sqrt(2), lengths[1])
    return DomainConstants(n0=n, original_area=float(area), average_edge_length=float(avg),
                           est_min_l=float(min_l), est_crit_l=float((max_l + 3 * min_l) / 4))
