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


def mesh_graph(quads: np.ndarray, vertex_xy: np.ndarray, domain_points: Sequence) -> dict:
    """The dictionary BoudaryEnv.write_2_file dumps (rl/boundary_env.py:648-669): every vertex with its coordinates
    and the ids it shares a segment with, in the order the reference created those segments -- the polygon's own edges
    first (connect_vertices, general/mesh.py:1926-1930), then, element by element, the quad edges that did not exist
    yet (Mesh.connect_vertices, general/components.py:832-837) -- and the elements as vertex ids."""
    n0 = len(domain_points)
    nv = len(vertex_xy)
    connected = [[] for _ in range(nv)]
    for i in range(n0):
        a, b = (i - 1) % n0, i
        connected[a].append(b)
        connected[b].append(a)
    for q in quads:
        for i in range(4):
            a, b = int(q[i]), int(q[i - 1])
            if b not in connected[a]:
                connected[a].append(b)
                connected[b].append(a)
    nodes = {}
    for v in range(nv):
        xy = domain_points[v] if v < n0 else (float(vertex_xy[v, 0]), float(vertex_xy[v, 1]))
        nodes[v] = {"coordinates": [xy[0], xy[1]], "connected": connected[v]}
    elements = {k: [int(v) for v in q] for k, q in enumerate(quads)}
    return {"nodes": nodes, "elements": elements}


def write_2_file(filename, quads, vertex_xy, domain_points) -> None:
    import json
    with open(filename, "w") as fw:
        json.dump(mesh_graph(quads, vertex_xy, domain_points), fw)
