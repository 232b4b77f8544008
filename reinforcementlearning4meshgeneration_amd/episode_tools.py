"""What the reference's callers do with ONE environment after (or during) an episode: read ``generated_meshes``, draw them
(``save_meshes``), score elements (``get_quality``), export them, dump samples.  Shared by the single-env drop-in
(``BoudaryEnv``) and by the per-env views of a vectorised batch (``MeshVecEnv.envs[k]``), which is what
``env.envs[0].save_meshes(..., meshes=env.envs[0].generated_meshes, ...)`` of the evaluation callback addresses
(rl/baselines/CustomizeCallback.py:131-133).

A class that mixes this in provides ``_vec`` (the MeshVecEnv), ``_k`` (the env index in it) and ``points`` (the domain as
handed to the env: its Python numbers decide how original coordinates print in the exports).
"""
from __future__ import annotations

import json

import numpy as np


class EpisodeTools:
    _k = 0

    # ------------------------------------------------------------------ the mesh of the running episode
    def get_elements(self):
        """(quads [n,4] vertex ids, vertices [m,2]) -- what write_generated_elements_2_file consumes."""
        return self._vec.get_elements(self._k)

    @property
    def generated_meshes(self):
        """List of quads, each a [4, 2] array of vertex coordinates in Mesh.vertices order (rl/boundary_env.py:192)."""
        quads, vxy = self._vec.get_elements(self._k)
        return [vxy[q] for q in quads]

    def write_generated_elements_2_file(self, filename):
        """Abaqus .inp of the current episode's mesh (general/mesh.py:1842-1864); byte-identical to the reference's file."""
        from .export import write_inp
        quads, vxy = self._vec.get_elements(self._k)
        write_inp(filename, quads, vxy, self.points)

    def write_2_file(self, filename):
        """JSON node / element dump (rl/boundary_env.py:648-669)."""
        from .export import write_2_file
        quads, vxy = self._vec.get_elements(self._k)
        write_2_file(filename, quads, vxy, self.points)

    def element_quality(self):
        """[n_elem, 8] float64: min / max corner angle (deg), scaled Jacobian, stretch, taper, robust, area, default
        -- Mesh.get_quality(type) of every generated element (general/components.py:863-950), computed on the GPU."""
        rec, _, cnt = self._vec.element_quality("current")
        return rec[self._k, :int(cnt[self._k])].cpu().numpy()

    # ------------------------------------------------------------------ get_quality / save_meshes / save_samples
    def get_quality(self, element, index=0):
        """MeshGeneration.get_quality(element, index), general/mesh.py:1728-1747 -- e.g. ``env.get_quality(env.generated_meshes[i],
        4)`` of rl/baselines/testbed.py:194-197.  element: a [4, 2] array (an entry of ``generated_meshes``) or an object
        with ``.vertices[k].x / .y``.  Computed on the device (meshenv_quad_quality) for index 0 'default', 1
        compute_element_quality, 3 'stretch', 4 'robust', 5 'strong'.  Index 2 / 6 add the boundary term of the ring right
        after the extraction: the reference evaluates them inside step() only (rl/boundary_env.py:219) and so does the
        step kernel -- asking for them afterwards raises ValueError."""
        from .plotting import element_xy
        if index in (2, 6):
            raise ValueError(f"get_quality(element, {index}) includes compute_ele_boundary_quality of the ring at the "
                             "moment of extraction; it is part of that step's reward and cannot be recomputed later")
        return float(self._vec.quad_quality(element_xy(element)[None], int(index))[0])

    def _quality_of(self, meshes, index):
        from .plotting import element_xy
        if index in (2, 6):
            raise ValueError(f"save_meshes(quality=True, type={index}): see get_quality")
        return self._vec.quad_quality(np.stack([element_xy(m) for m in meshes]), int(index))

    @property
    def last_generated_meshes(self):
        """`generated_meshes` of the last FINISHED episode (kept across auto-reset: the reference's callers read the finished
        mesh before their own reset, which an auto-resetting batch has already done for them)."""
        ep = self._vec.get_last_episode(self._k)
        return [ep["vertex_xy"][q] for q in ep["quads"]]

    def save_meshes(self, name, meshes, quality=False, indexing=False, type=0, dpi=300, style='k.-', *, which="current"):
        """MeshGeneration.save_meshes, general/mesh.py:1785-1792: a PNG of the domain with every element edge generated so
        far (the reference draws its whole vertex graph, whatever `meshes` holds) and a label on each element of
        `meshes`: its index (indexing), its get_quality(element, type) (quality), or both.  which="last" (extension, keyword
        only) draws the graph of the archived finished episode instead -- pair it with `last_generated_meshes` under
        auto-reset."""
        from .plotting import save_meshes
        if which == "last":
            ep = self._vec.get_last_episode(self._k)
            quads, vxy = ep["quads"], ep["vertex_xy"]
        elif which == "current":
            quads, vxy = self._vec.get_elements(self._k)
        else:
            raise ValueError("which must be 'current' or 'last'")
        save_meshes(name, len(self.points), quads, vxy, meshes, quality=quality, indexing=indexing, type=type, dpi=dpi,
                    style=style, quality_of=self._quality_of)

    @staticmethod
    def points_as_array(points):
        """general/mesh.py:1641-1646: [x0, y0, x1, y1, ...] of points given as objects with .x / .y or as (x, y) pairs."""
        flat = []
        for p in points:
            x, y = (p.x, p.y) if hasattr(p, "x") else (p[0], p[1])
            flat.append(float(x))
            flat.append(float(y))
        return flat

    def save_samples(self, file_name, res, _type=1):
        """MeshGeneration.save_samples, general/mesh.py:1634-1639: JSON dump of {'samples', 'output_types', 'outputs'};
        _type=1 flattens point lists first (points_as_array), _type=2 (what testbed.py:218-219 passes with the lists
        extract_samples_2 returns) writes them as they are."""
        if _type == 1:
            res['samples'] = [self.points_as_array(s) for s in res['samples']]
            res['outputs'] = [self.points_as_array(s) for s in res['outputs']]
        with open(file_name, 'w') as fw:
            json.dump(res, fw)

    # ------------------------------------------------------------------ state the reference's callers read
    @property
    def failed_num(self):
        return self._vec.get_state(self._k)["failed_num"]

    @property
    def current_area(self):
        return self._vec.get_state(self._k)["current_area"]

    @property
    def updated_boundary(self):
        """Current front as an [n, 2] array (ring order)."""
        return self._vec.get_state(self._k)["ring_xy"]
