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


def soup_scene(tmp_path, kind, scale=1.0, offset=(0.0, 0.0, 0.0), n=300, seed=7):
    """A triangle soup that stresses tie-breaking and degenerate input: `dup` lists a third of the faces twice -- the
    copy with OPPOSITE vertex normals, so the film shows which copy won --, `degenerate` has zero-area faces (two equal
    vertices, collinear, a point), `flat` puts every face in one plane (z-fighting), `plain` is none of those."""
    import numpy as np
    r = np.random.default_rng(seed)
    tri = r.uniform(-1, 1, (n, 1, 3)) + r.normal(scale=0.15, size=(n, 3, 3))
    nrm = np.tile(np.array([0.0, 0.0, 1.0]), (n, 1))
    if kind == "degenerate":
        tri[::5, 1] = tri[::5, 0]
        tri[1::5, 2] = (tri[1::5, 0] + tri[1::5, 1]) / 2
        tri[2::5] = tri[2::5, :1]
    elif kind == "dup":
        tri[n // 3:2 * (n // 3)] = tri[:n // 3]
        nrm[n // 3:2 * (n // 3)] = [0.6, 0.0, -0.8]
    elif kind == "flat":
        tri[:, :, 2] = 0.25
    tri = tri * scale + np.asarray(offset)
    with open(tmp_path / "soup.obj", "w") as f:
        for t in tri:
            for v in t:
                f.write("v %.9g %.9g %.9g\n" % tuple(v))
        for v in nrm:
            f.write("vn %.9g %.9g %.9g\n" % tuple(v))
        for i in range(n):
            f.write("f %d//%d %d//%d %d//%d\n" % (3 * i + 1, i + 1, 3 * i + 2, i + 1, 3 * i + 3, i + 1))
    ctr = np.asarray(offset, float)
    cam = ctr + np.array([0.3, 0.4, 4.0]) * scale
    (tmp_path / "soup.yaml").write_text(
        "film:\n    width: 48\n    height: 48\n    samples: 4\n    output: o.png\n"
        "camera:\n    position: [%.9g, %.9g, %.9g]\n    look_at: [%.9g, %.9g, %.9g]\n    up: [0, 1, 0]\n    fov: 40\n"
        "    aperture: 0\n    focal_distance: 4\n    background: [0.5, 0.6, 0.8]\n" % (tuple(cam) + tuple(ctr)) +
        "materials:\n  - name: m\n    type: metal\n    albedo: [0.8, 0.6, 0.5]\n    roughness: 0.2\n"
        "objects:\n  - type: mesh\n    path: soup.obj\n    material: m\n")
    return str(tmp_path / "soup.yaml"), ctr
