"""TEST INFRASTRUCTURE (container-only): known-answer vectors of the front smoother's three vertex constructions --
MeshGeneration.middle_vertex / side_vertex / indention_vertex (general/mesh.py:805-909) and smooth()'s
Mesh.estimate_4th_vertex (general/components.py:980-990) -- evaluated by the reference itself
on random and on axis-aligned inputs (the B == 0 / A == 0 branches).  -> tests/golden/front_constructions.npz"""
from __future__ import annotations

import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_harness as H  # noqa: E402


def main():
    Env, Vertex, Segment, B2 = H.load_reference()
    env = H.make_env(H.domain_points("boundary0"))
    rng = np.random.default_rng(2024)
    rows, outs, raised = [], [], []
    from mesh_rl.legacy.components_legacy import Mesh
    for which in (0, 1, 2, 3):
        for k in range(4000):
            v, a, b = rng.uniform(-3, 3, 2), rng.uniform(-3, 3, 2), rng.uniform(-3, 3, 2)
            kind = k % 8
            if kind == 5:   # B == 0: the second difference in y vanishes
                (b if which == 0 else b)[1] = a[1] if which != 2 else b[1]
                if which == 0: b[1] = a[1]
                elif which == 1: b[1] = a[1]
                else: a[1] = v[1]
            elif kind == 6:  # A == 0
                if which == 0: b[0] = a[0]
                elif which == 1: b[0] = a[0]
                else: a[0] = v[0]
            elif kind == 7:  # grid coordinates (boundary()-like)
                v, a, b = np.round(v), np.round(a), np.round(b)
            angle = float(rng.uniform(5, 175)) if which else float(rng.choice([45, 50, 60, 75, 90, 110, 130]))
            dist = float(rng.uniform(0.05, 2.5))
            if which == 3:   # estimate_4th_vertex: factor in the angle slot, suggest_dist (or -1 = None) in the dist slot
                angle = float(rng.choice([0.5, 0.7]))
                dist = dist if k % 2 else -1.0
            V, A, B = Vertex(float(v[0]), float(v[1])), Vertex(float(a[0]), float(a[1])), Vertex(float(b[0]), float(b[1]))
            try:
                if which == 0:
                    r = env.middle_vertex(V, A, B, angle)
                elif which == 1:
                    r = env.side_vertex(V, A, B, angle, dist)
                elif which == 2:
                    r = env.indention_vertex(V, A, B, angle, dist)
                else:
                    r = Mesh.estimate_4th_vertex(V, A, B, factor=angle, suggest_dist=dist if dist >= 0 else None)
                out, bad = (float(r.x), float(r.y)), 0
            except (ValueError, ZeroDivisionError):
                out, bad = (0.0, 0.0), 1
            rows.append([which, v[0], v[1], a[0], a[1], b[0], b[1], angle, dist])
            outs.append(out)
            raised.append(bad)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "front_constructions.npz"), inputs=np.array(rows, np.float64),
                        outputs=np.array(outs, np.float64), raised=np.array(raised, np.uint8))
    r = np.array(raised)
    print(f"{len(rows)} evaluations, {int(r.sum())} where the reference raises, NaN outputs {int(np.isnan(np.array(outs)).any(axis=1).sum())}")


if __name__ == "__main__":
    main()
