#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the CPU ORACLE (oracle/rt_oracle.c).

The reference (camms205/ray-tracing) holds no golden vectors, fixtures or tests and cannot be run
here, so these fixtures pin the oracle against itself over time ("parity unpinned" by the reference;
the hand-worked KATs of SURVEY.md section 4 in tests/test_oracle_kat.py are the independent pin).
Inputs (scene arrays, view) are stored with the outputs, so a fixture is self-contained data.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import oracle_py as orc  # noqa: E402
import ray_tracing_amd as rt  # noqa: E402


def save(name, app, W, H, flags, spp=0, seed=0):
    scene = app.prepare_meshinfo()
    m, p = app.camera_view()
    osc = orc.Scene(scene.triangles, scene.meshes, scene.vertices, scene.materials)
    view = orc.make_view(m, p, W, H)
    rgba, prim, t, rays = orc.render_primary(osc, view, flags, threads=4)
    out = dict(triangles=scene.triangles, meshes=scene.meshes, vertices=scene.vertices, materials=scene.materials,
               world_from_clip=m, world_position=p, size=np.array([W, H], np.uint32), flags=np.uint32(flags),
               rgba=rgba, prim=prim, t=t, rays=np.uint64(rays))
    if spp:
        acc, segs = orc.render_path(osc, view, spp, 4, seed=seed, threads=4)
        out.update(path_accum=acc, path_spp=np.uint32(spp), path_seed=np.uint32(seed), path_segments=np.uint64(segs))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "tris", len(scene.triangles), "rays", rays, "hit", float((prim != 0xFFFFFFFF).mean()))


def main():
    app = rt.App()
    app.spawn_default_scene()
    app.camera_set_target(64, 36)
    save("default_64x36_shadow", app, 64, 36, orc.FLAG_SHADOW, spp=2, seed=11)

    app = rt.App()
    # 8 overlapping ico(2) spheres on the corners of a cube + a rotated, scaled cuboid: exercises
    # inter-mesh occlusion, prepare_meshinfo's scale-ignoring AABB (culls part of the cuboid) and ties
    k = 0
    for x in (-1.5, 1.5):
        for y in (-1.5, 1.5):
            for z in (-1.5, 1.5):
                mat = app.add_material((0.2 + 0.1 * k, 1.0 - 0.1 * k, 0.5, 1.0))
                app.spawn(app.add_icosphere(1.7, 2), mat, translation=(x, y, z))
                k += 1
    mat = app.add_material((0.9, 0.9, 0.1, 1.0))
    app.spawn(app.add_cuboid(1.0, 1.0, 1.0), mat, scale=(3.0, 0.5, 0.5), rotation=(0.0, 0.38268343, 0.0, 0.92387953),
              translation=(0.0, 0.0, 3.5))
    app.camera_look_at((0.5, 1.0, 9.0))
    app.camera_set_target(48, 32)
    save("spheres8_48x32", app, 48, 32, 0)


if __name__ == "__main__":
    main()
