"""Scene generators shared by the CPU and GPU test modules."""


def chain_scene(tmp_path, n, ratio):
    """A mesh whose triangles grow geometrically along x: the SAH builder peels them off one by one, so the tree is about
    n / 3.7 levels deep (every other test mesh stays under 20)."""
    with open(tmp_path / "chain.obj", "w") as f:
        f.write("vn 0 0 1\n")
        for i in range(n):
            s = ratio ** i
            x = 2.0 * s
            f.write(f"v {x:.9g} {-0.5 * s:.9g} 0\nv {x + s:.9g} {-0.5 * s:.9g} 0\nv {x + 0.5 * s:.9g} {0.5 * s:.9g} {0.1 * s:.9g}\n")
            f.write(f"f {3 * i + 1}//1 {3 * i + 2}//1 {3 * i + 3}//1\n")
    (tmp_path / "s.yaml").write_text(
        "film:\n    width: 64\n    height: 48\n    samples: 4\n    output: o.png\n"
        "camera:\n    position: [6, 1, 14]\n    look_at: [6, 0, 0]\n    up: [0, 1, 0]\n    fov: 50\n    aperture: 0\n    focal_distance: 14\n"
        "    background: [0.5, 0.6, 0.8]\n"
        "materials:\n  - name: m\n    type: metal\n    albedo: [0.8, 0.6, 0.5]\n    roughness: 0.3\n"
        "  - name: g\n    type: lambertian\n    albedo: [0.5, 0.5, 0.5]\n"
        "objects:\n  - type: xz_rect\n    x: [-50, 50]\n    z: [-50, 50]\n    k: -1\n    material: g\n"
        "  - type: mesh\n    path: chain.obj\n    material: m\n")
    return str(tmp_path / "s.yaml")
