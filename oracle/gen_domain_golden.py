"""TEST INFRASTRUCTURE (container-only): golden vectors for the domain pipeline.

The reference's polygon generator and its density / orientation code live in scripts that cannot be imported
(`ui/GenerateRandomPolygon.py` opens a PIL window at import, `ui/tk-ui.py` builds a Tk GUI), so this script reads the
two files as text AT RUN TIME, takes the named function definitions out of their syntax trees and executes exactly
those -- nothing is copied into the repository:

  ui/GenerateRandomPolygon.py   generatePolygon (5-49), clip (52-60)        under random.seed(s)
  ui/tk-ui.py                   Density.calculate_density (252-276), Density.distance (248-251),
                                clockwise_angle (185-192), MeshFrame.check_clockwise (169-176)
                                (the methods are called with a plain namespace object standing in for `self`)

and writes tests/golden/domain_pipeline.json: inputs and the reference's outputs (data only).

usage: python oracle/gen_domain_golden.py
"""
from __future__ import annotations

import ast
import json
import math
import os
import random
import types

REFERENCE_ROOT = os.environ.get("MESHENV_REFERENCE_ROOT", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "domain_pipeline.json")


def functions_of(path, names):
    """name -> function object for the (possibly nested in a class) definitions `names` of the file, compiled on their
    own in a namespace that only holds math / random."""
    tree = ast.parse(open(path).read(), filename=path)
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
            found[node.name] = node
    missing = set(names) - set(found)
    if missing:
        raise SystemExit(f"{path}: no definition of {sorted(missing)}")
    ns = {"math": math, "random": random}
    mod = ast.Module(body=[found[n] for n in names], type_ignores=[])
    exec(compile(mod, path, "exec"), ns)
    return {n: ns[n] for n in names}


class _Entry:  # stands in for a tkinter Entry: .get() returns the typed text
    def __init__(self, text):
        self.text = text

    def get(self):
        return self.text


def run_density(F, points, base_length, densities):
    """Density.calculate_density(self, event) with a namespace `self` (the GUI fields it reads / writes)."""
    base_frame = types.SimpleNamespace(points=None, _create_circle=lambda *a, **k: None)
    self = types.SimpleNamespace(points=[tuple(p) for p in points], base_entry=_Entry(repr(base_length)),
                                 density_entries=[_Entry(repr(d)) for d in densities], base_frame=base_frame)
    self.distance = lambda a, b: F["distance"](self, a, b)
    # calculate_density calls the module-level clockwise_angle and print(); both are in its globals
    F["calculate_density"].__globals__["clockwise_angle"] = F["clockwise_angle"]
    F["calculate_density"].__globals__["print"] = lambda *a, **k: None
    F["calculate_density"](self, None)
    return [list(p) for p in base_frame.points]


def run_file_save_orientation(F, points):
    """The orientation part of MeshFrame.file_save (ui/tk-ui.py:84-101): reversed unless check_clockwise()."""
    self = types.SimpleNamespace(points=[tuple(p) for p in points])
    cw = F["check_clockwise"](self)
    return bool(cw), [list(p) for p in (self.points if cw else list(reversed(self.points)))]


def main():
    G = functions_of(os.path.join(REFERENCE_ROOT, "ui", "GenerateRandomPolygon.py"), ["clip", "generatePolygon"])
    G["generatePolygon"].__globals__["clip"] = G["clip"]
    F = functions_of(os.path.join(REFERENCE_ROOT, "ui", "tk-ui.py"),
                     ["clockwise_angle", "distance", "calculate_density", "check_clockwise"])
    out = {"generate_polygon": [], "calculate_density": [], "orientation": [], "pipeline": []}

    # (1) generatePolygon under random.seed(s): the call of ui/GenerateRandomPolygon.py:63 and config 5's sweep
    for seed, nv, irr, spk in [(0, 16, 0.55, 0.7), (1, 16, 0.55, 0.7), (2, 8, 0.55, 0.7), (3, 64, 0.55, 0.7),
                               (4, 33, 0.55, 0.7), (5, 12, 0.0, 0.0), (6, 20, 1.0, 1.0), (7, 24, 0.3, 0.2),
                               (12345, 40, 0.55, 0.7), (2 ** 40 + 17, 16, 0.55, 0.7), (999, 9, 2.0, -1.0)]:
        random.seed(seed)
        pts = G["generatePolygon"](ctrX=250, ctrY=250, aveRadius=100, irregularity=irr, spikeyness=spk, numVerts=nv)
        out["generate_polygon"].append(dict(seed=seed, num_verts=nv, irregularity=irr, spikeyness=spk,
                                            points=[list(p) for p in pts]))

    # (2) calculate_density: uniform and graded densities, integer and float corners, even/odd totals
    square = [(100, 100), (100, 400), (400, 400), (400, 100)]
    cases = [
        (square, 30.0, [1, 1, 1, 1]),
        (square, 45.0, [1, 1, 1, 1]),
        (square, 20.0, [1, 2, 1, 2]),
        (square, 25.0, [0.5, 1.5, 1.0, 2.0]),
        ([(33, 454), (705, 439), (716, 77), (363, 239), (44, 68)], 22.0, [1, 1, 0.6, 0.5, 1]),
        ([(50.5, 60.25), (300.75, 90.5), (280.0, 333.3), (120.1, 290.9), (20.0, 180.0)], 18.0, [1, 1.3, 0.8, 1.1, 0.9]),
    ]
    for seed in (0, 3, 12345):   # generated polygons as the UI would densify them (density 1 everywhere)
        random.seed(seed)
        nv = 8 + seed % 9
        pts = G["generatePolygon"](ctrX=250, ctrY=250, aveRadius=100, irregularity=0.55, spikeyness=0.7, numVerts=nv)
        if len(set(pts)) == len(pts):
            cases.append((pts, 45.0, [1] * nv))
    for pts, base, dens in cases:
        try:
            res = run_density(F, pts, base, [float(d) for d in dens])
        except ZeroDivisionError:
            res = None   # x == 0 on a short edge: the reference raises; the restatement must too
        out["calculate_density"].append(dict(points=[list(p) for p in pts], base_length=base,
                                             densities=[float(d) for d in dens], result=res))

    # (3) orientation normalisation on save
    for pts in (square, list(reversed(square)), [(0, 0), (4, 0), (4, 3)], [(0, 0), (4, 3), (4, 0)],
                [(10.5, 2.25), (3.0, 8.0), (-4.0, 1.0), (2.0, -6.5)]):
        cw, saved = run_file_save_orientation(F, pts)
        out["orientation"].append(dict(points=[list(p) for p in pts], check_clockwise=cw, saved=saved))

    # (4) the whole pipeline of a config-5 domain: generatePolygon -> density 1, base length b -> save -> /100
    for seed, nv, base in [(1000, 8, 12.0), (1001, 9, 10.0), (1002, 12, 8.0), (1003, 10, 9.0), (1005, 8, 11.0),
                           (1006, 16, 45.0), (1004, 24, 60.0)]:   # the last two: edges of 0.5-1.5 spacings, the reference raises
        random.seed(seed)
        raw = G["generatePolygon"](ctrX=250, ctrY=250, aveRadius=100, irregularity=0.55, spikeyness=0.7, numVerts=nv)
        keys = list(dict.fromkeys(raw))          # calculate_density keys a dict by point: duplicates collapse
        try:
            dense = run_density(F, raw, base, [1.0] * len(raw))
            _, saved = run_file_save_orientation(F, dense)
            ring = [[p[0] / 100, p[1] / 100] for p in saved]      # read_polygon, general/polygon.py:110-117
        except (ZeroDivisionError, IndexError):
            ring = None
        out["pipeline"].append(dict(seed=seed, num_verts=nv, base_length=base, raw=[list(p) for p in raw],
                                    distinct=len(keys), ring=ring))

    with open(OUT, "w") as f:
        json.dump(out, f)
    print(f"wrote {OUT}: {len(out['generate_polygon'])} polygons, {len(out['calculate_density'])} density cases, "
          f"{len(out['orientation'])} orientation cases, {len(out['pipeline'])} pipeline cases")


if __name__ == "__main__":
    main()
