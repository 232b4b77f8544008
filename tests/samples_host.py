"""TEST INFRASTRUCTURE: host restatement of MeshGeneration.extract_samples_2 (general/mesh.py:1438-1489), the
data-preparation step of the reference's ANN scripts (general/EBRD.py:414,579, general/post_processing.py:532).  The
product path is the device kernel (csrc/meshenv_samples.h, meshenv_extract_samples); this module is what the tests
compare it with on arbitrary meshes, and is itself pinned on the lists the reference returned
(tests/test_samples_cpu.py, tests/golden/samples_*.npz).

It consumes what the device produces, `MeshVecEnv.get_elements(env)` / `get_last_episode(env)` (elements as global
vertex ids, the vertex table, domain ring first), and rebuilds the `Vertex.segments` neighbour lists the way
`csrc/meshenv_smooth.h` does: the domain ring's own segments (general/mesh.py:1926-1930), then Mesh.connect_vertices
(general/components.py:832-837) of every element in order.

Every element of sufficient quality contributes, for each of its four corners as reference point rp (left / right
neighbour l_p / r_p, opposite corner = target): all neighbour paths of length n_neighbor leaving l_p and r_p (get_nodes,
:1422-1436), the radius neighbours of rp in n_radius angular sectors (get_radius_neighbors, :1601-1632), and from their
product the samples [distance / (base_length * radius), clockwise angle] per point, the target in the same frame, and a
type (1 / 0 / 0.5: the target lies on the right path / the left path / neither)."""
from __future__ import annotations

import itertools
import math
from typing import List, Sequence, Tuple

import numpy as np

Point = Tuple[float, float]


def _dist(a: Point, b: Point) -> float:
    return math.sqrt((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2)          # Point2D.distance_to, components.py:17-18


def _cw(s: Point, p1: Point, p2: Point) -> float:
    """Vertex.to_find_clockwise_angle, general/components.py:91-100."""
    v1x, v1y = p1[0] - s[0], p1[1] - s[1]
    v2x, v2y = p2[0] - s[0], p2[1] - s[1]
    theta = -math.atan2(v1x * v2y - v1y * v2x, v1x * v2x + v1y * v2y)
    return round(theta, 4) if math.copysign(1, theta) >= 0 else round(2 * math.pi + theta, 4)


def segment_lists(quads: np.ndarray, n_vert: int, n0: int) -> List[List[int]]:
    """get_connected_vertices() of every vertex, in the reference's order."""
    adj: List[List[int]] = [[] for _ in range(n_vert)]
    for i in range(n0):
        a = (i - 1) % n0
        adj[a].append(i)
        adj[i].append(a)
    for q in quads:
        for i in range(4):
            a, b = int(q[i]), int(q[i - 1])
            if b not in adj[a]:
                adj[a].append(b)
                adj[b].append(a)
    return adj


def _quality_3(p: Sequence[Point]) -> Tuple[float, float]:
    """Mesh.get_quality_3, components.py:952-972 (with compute_area, :935-950)."""
    edges = [_dist(p[i], p[i - 1]) for i in range(4)]
    area = 0.5 * edges[0] * edges[1] * math.sin(_cw(p[0], p[1], p[3])) + \
        0.5 * edges[2] * edges[3] * math.sin(_cw(p[2], p[3], p[1]))
    if area <= 0:
        q1 = 0
    else:
        product = 1
        for e in edges:
            product *= math.pow(e / math.sqrt(area), 1 if math.sqrt(area) - e > 0 else -1)
        q1 = math.pow(product, 1 / 4)
    angle_product = 1
    for i in range(4):
        angle_product *= 1 - (math.fabs(math.degrees(_cw(p[i], p[(i + 1) % 4], p[i - 1])) - 90) / 90)
    q2 = 0 if angle_product < 0 else math.pow(angle_product, 1 / 4)
    return q1, q2


def element_quality(p: Sequence[Point], index: int) -> float:
    """MeshGeneration.get_quality(element, index), general/mesh.py:1728-1747, for the two indices the callers use:
    1 = compute_element_quality (:1714-1726), 5 = Mesh.get_quality('strong') (components.py:907-930)."""
    q1, q2 = _quality_3(p)
    if index == 1:
        return math.pow(q1 * q2, 1 / 2)
    if index == 5:
        angles = [math.fabs(_cw(p[i], p[(i + 1) % 4], p[i - 1])) for i in range(4)]
        return math.sqrt(q1 * (min(angles) / max(angles)))
    raise NotImplementedError("extract_samples_2 is called with index 1 or 5 in the reference")


def neighbour_paths(adj, root: int, excluded: Sequence[int], N: int) -> List[List[int]]:
    """The paths of N vertices that MeshGeneration.get_nodes (general/mesh.py:1422-1436) emits from `root`, in its order,
    written as plain loops (N <= 3, the values of the reference's callers) -- the formulation csrc/meshenv_samples.h uses.

    The reference keeps ONE shared list: a node is appended while the list is shorter than N and written at its depth's
    position afterwards.  With N = 3 that matters exactly once: when the first second-level node a0 has no admissible
    child, the list is still [root, a0] when the next node a1 arrives, so a1 lands BEHIND a0 -- its children are filtered
    against [root, a0] (not a1) and come out as [root, a0, child].  From then on the list is full and positions are right."""
    assert 1 <= N <= 3
    if N == 1:
        return [[root]]
    level1 = [a for a in adj[root] if a not in excluded and a != root]
    if N == 2:
        return [[root, a] for a in level1]
    out: List[List[int]] = []
    first_dead = False
    for k, a in enumerate(level1):
        behind = k == 1 and first_dead
        shown, skip = (level1[0], level1[0]) if behind else (a, a)
        kids = [b for b in adj[a] if b not in excluded and b != root and b != skip]
        if k == 0:
            first_dead = not kids
        out.extend([root, shown, b] for b in kids)
    return out


def _radius_neighbors(xy, base: int, start: int, end: int, exclusion: Sequence[int], radius: float, N: int):
    """general/mesh.py:1601-1632.  Entries are vertex ids, or coordinate pairs for the synthetic sector points."""
    pb, ps, pe = xy[base], xy[start], xy[end]

    def sector(start_angle, end_angle):
        base_length = radius * (0.5 * _dist(pb, ps) + 0.5 * _dist(pb, pe))
        inside = [v for v in range(len(xy)) if start_angle < _cw(pb, ps, xy[v]) < end_angle]   # get_points_within_angle
        dists = sorted(((v, _dist(pb, xy[v])) for v in inside if v != base), key=lambda t: t[1])   # compute_dist
        close = []
        for v, d in dists:                                                                      # get_closet_points
            if v not in exclusion and d <= base_length:
                close.append(v)
            if d > base_length:
                break
        a = _cw(pb, ps, (pb[0] + 1, pb[1]))
        mid = a - (start_angle + end_angle) / 2
        close.append((pb[0] + base_length * math.cos(mid), pb[1] + base_length * math.sin(mid)))
        return close

    angle = _cw(pb, ps, pe)
    angles = [i * angle / N for i in range(N + 1)]
    neighbors = [sector(angles[i - 1], angles[i]) for i in range(1, N + 1)]
    return list(itertools.product(*reversed(neighbors)))


def extract_samples_2(quads, vertex_xy, n0: int, n_neighbor: int, n_radius: int, radius: float, index: int = 1,
                      quality_threshold: float = 0.7):
    """(all_samples, types, outputs) of MeshGeneration.extract_samples_2(meshes, n_neighbor, n_radius, radius, index,
    quality_threshold) for the mesh `quads` [n_elem, 4] (global vertex ids) over `vertex_xy` [n_vert, 2] whose first n0
    rows are the domain ring."""
    quads = np.asarray(quads, np.int64).reshape(-1, 4)
    xy = [(float(x), float(y)) for x, y in np.asarray(vertex_xy, np.float64).reshape(-1, 2)]
    adj = segment_lists(quads, len(xy), n0)
    two_pi = round(2 * math.pi, 4)

    def P(v):
        return xy[v] if isinstance(v, (int, np.integer)) else v

    all_samples, outputs, types = [], [], []
    for q in quads:
        ev = [int(v) for v in q]
        if element_quality([xy[v] for v in ev], index) < quality_threshold:
            continue
        for i in range(4):
            rp, l_p, r_p, target = ev[i], ev[(i + 1) % 4], ev[i - 1], ev[i - 2]
            l_paths = neighbour_paths(adj, l_p, [rp, r_p], n_neighbor)
            r_paths = neighbour_paths(adj, r_p, [rp, l_p], n_neighbor)
            fans = _radius_neighbors(xy, rp, l_p, r_p, [rp, l_p, r_p, target], radius, n_radius)
            for rr, mm, ll in itertools.product([p for p in r_paths if len(p) == n_neighbor], fans,
                                                [p for p in l_paths if len(p) == n_neighbor]):
                if target in rr and target in ll:
                    continue
                base_length = (_dist(xy[rp], xy[rr[0]]) +
                               sum(_dist(xy[rr[j]], xy[rr[j - 1]]) for j in range(1, len(rr))) +
                               sum(_dist(xy[ll[j]], xy[ll[j - 1]]) for j in range(1, len(ll))) +
                               _dist(xy[rp], xy[ll[0]])) / (2 * n_neighbor)
                sample = []
                for p in itertools.chain(rr, mm, reversed(ll)):
                    sample.extend([_dist(xy[rp], P(p)) / (base_length * radius), _cw(xy[rp], P(p), xy[r_p]) % two_pi])
                outputs.append([_dist(xy[rp], xy[target]) / (base_length * radius), _cw(xy[rp], xy[target], xy[r_p]) % two_pi])
                types.append([1] if target in rr else ([0] if target in ll else [0.5]))
                all_samples.append(sample)
    return all_samples, types, outputs
