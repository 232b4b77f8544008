"""Mesh export of a finished episode: the Abaqus .inp file of MeshGeneration.write_generated_elements_2_file
(general/mesh.py:1842-1864) -- *NODE list in order of first use, one B21 element per boundary edge of the original
polygon, then the S4R quadrilaterals."""
from __future__ import annotations

from typing import Sequence

import numpy as np


def inp_text(quads: np.ndarray, vertex_xy: np.ndarray, domain_points: Sequence) -> str:
    """quads [n_elem, 4] global vertex ids, vertex_xy [n_vert, 2] (MeshVecEnv.get_elements); domain_points = the
    polygon as handed to the env (its Python numbers decide how original coordinates print: `0` vs `0.0`)."""
    n0 = len(domain_points)
    if len(quads) == 0:
        raise ValueError("There are no elements generated!")
    order, index = [], {}
    for q in quads:                       # nodes in order of first appearance (general/mesh.py:1847-1849)
        for v in q:
            v = int(v)
            if v not in index:
                index[v] = len(order)
                order.append(v)

    def coords(v):
        if v < n0:
            return domain_points[v][0], domain_points[v][1]
        return float(vertex_xy[v, 0]), float(vertex_xy[v, 1])    # np.float64 prints like a Python float

    out = ["*NODE, NSET=ALLNODES\n"]
    for k, v in enumerate(order):
        x, y = coords(v)
        out.append(f"{k + 1}, {x}, {y}\n")
    i = 0
    for i in range(1, n0):
        if (i - 1) not in index or i not in index:
            raise ValueError(f"boundary vertex {i - 1 if (i - 1) not in index else i} is not part of any element "
                             "(the reference's writer fails on an unfinished mesh in the same way)")
        out.append(f"*ELEMENT, TYPE=B21, ELSET=EB{i}\n {i + 1}, {index[i - 1] + 1}, {index[i] + 1}\n")
    out.append(f"*ELEMENT, TYPE=S4R, ELSET=EB{i + 1} \n")
    for k, q in enumerate(quads):
        out.append(f"{k + 1}, {index[int(q[0])] + 1}, {index[int(q[1])] + 1}, {index[int(q[2])] + 1}, {index[int(q[3])] + 1}\n")
    return "".join(out)


def write_inp(filename, quads, vertex_xy, domain_points) -> None:
    with open(filename, "w") as fw:
        fw.write(inp_text(quads, vertex_xy, domain_points))
