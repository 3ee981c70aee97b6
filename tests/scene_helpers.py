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


def _v3(v): return "[%.9g, %.9g, %.9g]" % tuple(v)
def random_world(tmp_path, seed, extreme, meshes=False, images=False, bare_glass=False):
    """A random world of analytic primitives (spheres, rects, boxes, media, free triangles) on a 0.5 lattice, so that
    coplanar / coincident surfaces are common, with random transform chains and materials and a few objects listed twice
    with another material (exact ties between primitives).  `extreme` adds degenerate parameters: zero / negative radii and
    box dimensions, reversed or empty rect ranges, negative and huge scales, ior 1 / 0.5 / 1e-3 / 50, roughness 2.5,
    medium densities 0 / 1e-6 / 1e4 / -1.  `meshes` inserts 1..3 triangle meshes (a soup, a smooth-shaded sphere, a lattice-
    aligned grid that is coplanar with rects and box faces) at random places of the object list, under wrapper chains of their
    own: the wavefront pipeline walks the list mesh by mesh, analytic primitives in between (k_wf_pre).  `images` adds a
    7x5 PNG image texture (as albedo and as roughness) and a 16x8 Radiance environment map as background, so texel fetches
    see the (u, v) every primitive kind produces -- including the NaN and out-of-range ones of degenerate primitives.
    `bare_glass` (with `meshes`): half of the meshes stand in the list without a wrapper and get a dielectric material -- their
    hits inherit hitRecord::frontFace from the previous object of the walk (hrt_device.h WorldHit)."""
    import re
    import numpy as np
    d = tmp_path
    r=np.random.default_rng(seed)
    mats=[]
    def col(): return r.uniform(0,1,3)
    names=[]
    for i in range(8):
        k=r.integers(0,7)
        n="m%d"%i; names.append(n)
        if k==0: mats.append(f"  - name: {n}\n    type: lambertian\n    albedo: {_v3(col())}\n")
        elif k==1: mats.append(f"  - name: {n}\n    type: metal\n    albedo: {_v3(col())}\n    roughness: %.6g\n"%(r.choice([0,0.1,0.5,1.0,2.5]) if extreme else r.uniform(0,0.6)))
        elif k==2: mats.append(f"  - name: {n}\n    type: dielectric\n    ior: %.6g\n"%(r.choice([1.0,1.5,0.5,2.4,1e-3,50]) if extreme else r.uniform(1.1,2)))
        elif k==3: mats.append(f"  - name: {n}\n    type: diffuse_light\n    albedo: {_v3(col())}\n    strength: %.6g\n"%r.uniform(0.5,8))
        elif k==4: mats.append(f"  - name: {n}\n    type: pbr\n    albedo: {_v3(col())}\n    metalness: %.6g\n    roughness: %.6g\n"%(r.uniform(0,1),r.uniform(0,1)))
        elif k==5: mats.append(f"  - name: {n}\n    type: uv_test\n")
        else: mats.append(f"  - name: {n}\n    type: lambertian\n    albedo: chk\n")
    objs=[]
    grid=lambda: np.round(r.uniform(-2,2,3)*2)/2   # coincidences on a 0.5 lattice
    def xf():
        if r.random()<0.5: return ""
        s="    transform:\n"
        if r.random()<0.4: s+="        rotate_y: %.6g\n"%r.choice([0,90,45,r.uniform(-180,180)])
        if r.random()<0.3: s+="        rotate: %s\n"%_v3(r.uniform(-90,90,3))
        if r.random()<0.4: s+="        scale: %s\n"%_v3(r.choice([0.5,1,2,1.5],3) if not extreme else r.choice([0.5,1,2,-1,1e-3,100],3))
        if r.random()<0.6: s+="        translate: %s\n"%_v3(grid())
        return s if s!="    transform:\n" else ""
    for i in range(r.integers(3,14)):
        k=r.integers(0,8); m=names[r.integers(0,8)]
        if k==0:
            rad=r.choice([0.5,1.0,0.25]) if not extreme else r.choice([0.5,1.0,0,-0.5,1e-4,30])
            objs.append(f"  - type: sphere\n    center: {_v3(grid())}\n    radius: %.6g\n    material: {m}\n"%rad+xf())
        elif k in (1,2,3):
            t=["xy_rect","xz_rect","yz_rect"][k-1]; ax={"xy_rect":"xy","xz_rect":"xz","yz_rect":"yz"}[t]
            a=np.sort(np.round(r.uniform(-3,3,2)*2)/2); b=np.sort(np.round(r.uniform(-3,3,2)*2)/2)
            if extreme and r.random()<0.3: a=a[::-1]
            if extreme and r.random()<0.2: b[1]=b[0]
            objs.append(f"  - type: {t}\n    {ax[0]}: [%.6g, %.6g]\n    {ax[1]}: [%.6g, %.6g]\n    k: %.6g\n    material: {m}\n"%(a[0],a[1],b[0],b[1],np.round(r.uniform(-2,2)*2)/2)+xf())
        elif k==4:
            c=grid(); dm=r.choice([0.5,1,2],3) if not extreme else r.choice([0.5,1,0,-1,2],3)
            objs.append(f"  - type: box\n    center: {_v3(c)}\n    dimensions: {_v3(dm)}\n    material: {m}\n"+xf())
        elif k==5:
            lo=grid(); hi=lo+r.choice([0.5,1,1.5],3)
            objs.append(f"  - type: box\n    min: {_v3(lo)}\n    max: {_v3(hi)}\n    material: {m}\n"+xf())
        elif k==6:
            dens=r.uniform(0.2,3) if not extreme else r.choice([0,1e-6,1,1e4,-1])
            if r.random()<0.5: bnd=f"        type: sphere\n        center: {_v3(grid())}\n        radius: %.6g\n"%r.choice([0.5,1,1.5])
            else: bnd=f"        type: box\n        center: {_v3(grid())}\n        dimensions: {_v3(r.choice([1,2],3))}\n"
            objs.append(f"  - type: constant_medium\n    boundary:\n{bnd}    density: %.6g\n    colour: {_v3(col())}\n"%dens+xf())
        else:
            objs.append(f"  - type: triangle\n    v0: {_v3(grid())}\n    v1: {_v3(grid())}\n    v2: {_v3(grid())}\n    material: {m}\n"+xf())
    # duplicates: repeat some objects with another material (exact ties at world level)
    for j in range(r.integers(0,3)):
        o=objs[r.integers(0,len(objs))]
        if "material:" in o:
            objs.insert(r.integers(0,len(objs)+1), re.sub(r"material: m\d", "material: "+names[r.integers(0,8)], o))
    if meshes:
        rm = np.random.default_rng(10**6 + seed)          # its own stream: the analytic part of a seed stays what it was
        def xfm():
            t = "    transform:\n"
            if rm.random() < 0.4: t += "        rotate_y: %.6g\n" % rm.choice([90, 45, rm.uniform(-180, 180)])
            if rm.random() < 0.4: t += "        rotate: %s\n" % _v3(rm.uniform(-90, 90, 3))
            if rm.random() < 0.4: t += "        scale: %s\n" % _v3(rm.choice([0.5, 1, 2, 1.5], 3))
            if rm.random() < 0.7: t += "        translate: %s\n" % _v3(np.round(rm.uniform(-2, 2, 3) * 2) / 2)
            return t if t != "    transform:\n" else ""
        for i in range(int(rm.integers(1, 4))):
            kind = int(rm.integers(0, 3))
            with open(d / ("fmesh%d.obj" % i), "w") as f:
                if kind == 0:      # soup
                    n = int(rm.integers(1, 120))
                    tri = rm.uniform(-1, 1, (n, 1, 3)) + rm.normal(scale=0.3, size=(n, 3, 3))
                    f.write("vn 0 1 0\n")
                    for t in tri:
                        for v in t: f.write("v %.9g %.9g %.9g\n" % tuple(v))
                    for k in range(n): f.write("f %d//1 %d//1 %d//1\n" % (3 * k + 1, 3 * k + 2, 3 * k + 3))
                elif kind == 1:    # smooth sphere
                    nu, nv = int(rm.integers(3, 12)), int(rm.integers(2, 8))
                    rad = float(rm.choice([0.5, 1.0]))
                    for a in range(nv + 1):
                        for b in range(nu):
                            th, ph = np.pi * a / nv, 2 * np.pi * b / nu
                            nn = np.array([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)])
                            f.write("v %.9g %.9g %.9g\nvn %.9g %.9g %.9g\n" % (tuple(nn * rad) + tuple(nn)))
                    for a in range(nv):
                        for b in range(nu):
                            i0, i1 = a * nu + b + 1, a * nu + (b + 1) % nu + 1
                            j0, j1 = i0 + nu, i1 + nu
                            f.write("f %d//%d %d//%d %d//%d\nf %d//%d %d//%d %d//%d\n" % (i0, i0, j0, j0, j1, j1, i0, i0, j1, j1, i1, i1))
                else:              # lattice-aligned grid in the plane z = 0
                    g = int(rm.integers(1, 6))
                    f.write("vn 0 0 1\n")
                    for a in range(g + 1):
                        for b in range(g + 1): f.write("v %.9g %.9g 0\n" % (0.5 * b - 1, 0.5 * a - 1))
                    for a in range(g):
                        for b in range(g):
                            i0 = a * (g + 1) + b + 1
                            f.write("f %d//1 %d//1 %d//1\nf %d//1 %d//1 %d//1\n" % (i0, i0 + 1, i0 + g + 2, i0, i0 + g + 2, i0 + g + 1))
            mesh_obj = "  - type: mesh\n    path: fmesh%d.obj\n    material: %s\n" % (i, names[int(rm.integers(0, 8))]) + xfm()
            if bare_glass and np.random.default_rng(3 * 10**6 + 7 * seed + i).random() < 0.5:
                mesh_obj = "  - type: mesh\n    path: fmesh%d.obj\n    material: bareglass\n" % i
            objs.insert(int(rm.integers(0, len(objs) + 1)), mesh_obj)
    cam=r.uniform(-1,1,3)*np.array([3,2,1])+np.array([0,1,7])
    y=("film:\n    width: 40\n    height: 40\n    samples: 4\n    output: o.png\n"
       f"camera:\n    position: {_v3(cam)}\n    look_at: [0, 0, 0]\n    up: [0, 1, 0]\n    fov: 45\n    aperture: %.6g\n    focal_distance: 7\n    background: {_v3(col())}\n"%r.choice([0,0.1])+
       "textures:\n  - name: chk\n    type: checkered\n    even: [0.9, 0.9, 0.9]\n    odd: [0.1, 0.3, 0.1]\n"
       "materials:\n"+"".join(mats)+("  - name: bareglass\n    type: dielectric\n    ior: 1.5\n" if bare_glass else "")+"objects:\n"+"".join(objs))
    if images:
        from hobbyraytracer_amd import api
        ri = np.random.default_rng(2 * 10**6 + seed)
        api.write_image(str(d / "ftex.png"), ri.integers(0, 256, (5, 7, 3)).astype(np.uint8))
        api.write_hall_hdr(str(d / "fenv.hdr"), 16, 8)
        y = y.replace("textures:\n", "textures:\n  - name: img\n    type: image\n    path: ftex.png\n  - name: env\n    type: environment\n    path: fenv.hdr\n", 1)
        y = y.replace("albedo: chk", "albedo: img")
        lines = y.split("\n")
        kind = ""
        for k, line in enumerate(lines):
            if line.startswith("    type: "): kind = line[10:]
            if line.startswith("    albedo: [") and kind in ("lambertian", "metal", "diffuse_light") and ri.random() < 0.4: lines[k] = "    albedo: img"
            elif line.startswith("    roughness: ") and kind == "metal" and ri.random() < 0.5: lines[k] = "    roughness: img"
            elif line.startswith("    background: [") and ri.random() < 0.6: lines[k] = "    background: env"
        y = "\n".join(lines)
    open(d/"f.yaml","w").write(y)
    return str(d/"f.yaml")


def wrapper_chain_scene(tmp_path, chain):
    """One object of every kind (sphere, box, rect, free triangle, medium, mesh), each under the SAME wrapper chain:
    `chain` is a subset of "YQST" (rotate_y, rotate, scale, translate; the loader nests them in that order, scene.cpp:334-354),
    so every chain length 0..4 = HRT_MAX_XFORMS can be asked for."""
    xf = {"Y": "        rotate_y: 35\n", "Q": "        rotate: [-53.4, -38.9, -33.5]\n", "S": "        scale: [1, 1.5, 0.75]\n",
          "T": "        translate: [0.25, -0.5, 0.5]\n"}
    t = ("    transform:\n" + "".join(xf[c] for c in chain)) if chain else ""
    with open(tmp_path / "quad.obj", "w") as f:
        f.write("v -0.6 -0.6 0\nv 0.6 -0.6 0\nv 0.6 0.6 0.2\nv -0.6 0.6 0\nvn 0 0 1\nf 1//1 2//1 3//1\nf 1//1 3//1 4//1\n")
    (tmp_path / "chain.yaml").write_text(
        "film:\n    width: 48\n    height: 48\n    samples: 4\n    output: o.png\n"
        "camera:\n    position: [0.5, 1.5, 9]\n    look_at: [0, 0, 0]\n    up: [0, 1, 0]\n    fov: 45\n    aperture: 0\n"
        "    focal_distance: 9\n    background: [0.6, 0.7, 0.9]\n"
        "materials:\n  - name: a\n    type: lambertian\n    albedo: [0.8, 0.4, 0.3]\n  - name: b\n    type: metal\n    albedo: [0.8, 0.8, 0.7]\n"
        "    roughness: 0.1\n  - name: c\n    type: uv_test\n  - name: g\n    type: dielectric\n    ior: 1.5\n"
        "objects:\n"
        "  - type: sphere\n    center: [-2, 1.2, 0]\n    radius: 0.7\n    material: g\n" + t +
        "  - type: box\n    center: [0, 1.2, 0]\n    dimensions: [1, 1, 1]\n    material: c\n" + t +
        "  - type: xy_rect\n    x: [1.4, 2.6]\n    y: [0.6, 1.8]\n    k: 0\n    material: b\n" + t +
        "  - type: triangle\n    v0: [-2.6, -1.6, 0]\n    v1: [-1.4, -1.6, 0.3]\n    v2: [-2, -0.4, 0]\n    material: a\n" + t +
        "  - type: constant_medium\n    boundary:\n        type: sphere\n        center: [0, -1, 0]\n        radius: 0.7\n    density: 1.5\n"
        "    colour: [0.2, 0.8, 0.3]\n" + t +
        "  - type: mesh\n    path: quad.obj\n    material: b\n" + ("    transform:\n" + "".join(xf[c] for c in chain if c != "T") + "        translate: [2, -1, 0]\n") +
        "  - type: xz_rect\n    x: [-6, 6]\n    z: [-6, 6]\n    k: -2.2\n    material: a\n")
    return str(tmp_path / "chain.yaml")


def films_equal(a, b):
    """Bit equality of two linear films, with NaN == NaN: a NaN pixel (a path that met a degenerate primitive) may carry any
    payload or sign, and Film::tonemap (film.cpp:35-37) scrubs every NaN to 0 before anything is shown."""
    import numpy as np
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


def nan_ray_scene(tmp_path):
    """A free triangle with two equal vertices, listed BEFORE a mesh and a light.  Triangle::hit (triangle.cpp:4-40)
    normalises a zero edge: every comparison on the NaNs is false and the triangle "hits" every ray with t = NaN, so the
    objects after it are asked with t_max = NaN, and the path goes on from a NaN point: the NaN ray passes every box of the
    reference's tree (aabb.h:29-35 on NaN) and is "hit" by the mesh again."""
    with open(tmp_path / "quad.obj", "w") as f:
        f.write("v -1 -1 0\nv 1 -1 0\nv 1 1 0.3\nv -1 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\nf 1//1 3//1 4//1\n")
    (tmp_path / "nan.yaml").write_text(
        "film:\n    width: 32\n    height: 32\n    samples: 4\n    output: o.png\n"
        "camera:\n    position: [0.3, 0.8, 6]\n    look_at: [0, 0, 0]\n    up: [0, 1, 0]\n    fov: 45\n    aperture: 0\n"
        "    focal_distance: 6\n    background: [0.6, 0.7, 0.9]\n"
        "materials:\n  - name: a\n    type: lambertian\n    albedo: [0.8, 0.4, 0.3]\n  - name: lamp\n    type: diffuse_light\n"
        "    albedo: [1, 0.9, 0.8]\n    strength: 3\n  - name: b\n    type: metal\n    albedo: [0.8, 0.8, 0.7]\n    roughness: 0.1\n"
        "objects:\n"
        "  - type: sphere\n    center: [-1.5, 0, 0]\n    radius: 0.6\n    material: b\n"
        "  - type: triangle\n    v0: [1.5, 0.5, 1.5]\n    v1: [1.5, 0.5, 1.5]\n    v2: [1, 2, -0.5]\n    material: a\n"
        "  - type: mesh\n    path: quad.obj\n    material: lamp\n    transform:\n        rotate_y: 30\n"
        "  - type: xz_rect\n    x: [-4, 4]\n    z: [-4, 4]\n    k: -1.2\n    material: a\n"
        "  - type: mesh\n    path: quad.obj\n    material: b\n    transform:\n        translate: [0, 0, -2]\n")
    return str(tmp_path / "nan.yaml")


def many_meshes_scene(tmp_path, n_mesh, seed=None):
    """n_mesh instances of two OBJ files (a small teapot, a quad) under rotate_y + translate, a sphere after each, a floor:
    more meshes than k_wf_tail takes (HRT_TAIL_MAX_MESHES = 4), glass, metal and diffuse.  The teapot's shared edges make
    1-ulp near-ties between neighbouring triangles likely: the 5-mesh scene was the first to show one whose winner depends
    on the visiting order (hrt_device.h trav_result)."""
    import numpy as np
    from hobbyraytracer_amd import api
    api.write_teapot_obj(str(tmp_path / "teapot.obj"), 0.1)
    with open(tmp_path / "quad.obj", "w") as f:
        f.write("v -0.6 -0.6 0\nv 0.6 -0.6 0\nv 0.6 0.6 0.2\nv -0.6 0.6 0\nvn 0 0 1\nf 1//1 2//1 3//1\nf 1//1 3//1 4//1\n")
    r = np.random.default_rng(n_mesh if seed is None else seed)
    objs = ""
    for i in range(n_mesh):
        pos = r.uniform(-2, 2, 3)
        objs += "  - type: mesh\n    path: %s\n    material: %s\n    transform:\n        rotate_y: %.3f\n        translate: [%.3f, %.3f, %.3f]\n" % (
            "teapot.obj" if i % 3 == 0 else "quad.obj", ["a", "b", "g"][i % 3], r.uniform(0, 360), pos[0], pos[1] * 0.5, pos[2])
        objs += "  - type: sphere\n    center: [%.3f, %.3f, %.3f]\n    radius: 0.3\n    material: %s\n" % (pos[0] + 0.8, pos[1] * 0.5 + 0.5, pos[2], ["b", "g", "a"][i % 3])
    (tmp_path / "many.yaml").write_text(
        "film:\n    width: 48\n    height: 48\n    samples: 4\n    output: o.png\n"
        "camera:\n    position: [0.5, 1.5, 9]\n    look_at: [0, 0, 0]\n    up: [0, 1, 0]\n    fov: 45\n    aperture: 0\n    focal_distance: 9\n"
        "    background: [0.6, 0.7, 0.9]\n"
        "materials:\n  - name: a\n    type: lambertian\n    albedo: [0.8, 0.4, 0.3]\n  - name: b\n    type: metal\n    albedo: [0.8, 0.8, 0.7]\n"
        "    roughness: 0.1\n  - name: g\n    type: dielectric\n    ior: 1.5\n"
        "objects:\n" + objs + "  - type: xz_rect\n    x: [-6, 6]\n    z: [-6, 6]\n    k: -1.5\n    material: a\n")
    return str(tmp_path / "many.yaml")


def stale_front_face_scene(tmp_path, enclosed):
    """Glass meshes that stand in the world list WITHOUT a wrapper: ITriangle::hit never writes hitRecord::frontFace
    (triangle.cpp:118-128), so their hits carry the flag of the previous successful object of HittableList::hit's walk
    (hittableList.cpp:6-16) -- here, with `enclosed`, a huge sphere listed first and always met from INSIDE (frontFace false:
    Dielectric::scatter then refracts with ir instead of 1/ir, material.h:207); without it walls, a sphere and rectangles met
    from either side, or nothing at all (the flag's initial value).  Two such meshes in a row (the second one's hit inherits
    through the first's), a glass mesh WITH a wrapper (which writes the flag itself, translate.cpp:16) and objects behind them."""
    from hobbyraytracer_amd import api
    api.write_teapot_obj(str(tmp_path / "teapot.obj"), 0.1)
    with open(tmp_path / "quad.obj", "w") as f:
        f.write("v -1.2 -0.2 1.4\nv 1.2 -0.2 1.4\nv 1.2 1.6 1.1\nv -1.2 1.6 1.4\nvn 0 0 1\nf 1//1 2//1 3//1\nf 1//1 3//1 4//1\n")
    objs = ""
    if enclosed:
        objs += "  - type: sphere\n    center: [0, 0, 0]\n    radius: 30\n    material: sky\n"
    objs += ("  - type: xy_rect\n    x: [-4, 4]\n    y: [-2, 4]\n    k: -3\n    material: a\n"
             "  - type: sphere\n    center: [1.2, 0.4, -1.5]\n    radius: 0.7\n    material: b\n"
             "  - type: yz_rect\n    y: [-2, 4]\n    z: [-3, 3]\n    k: -3\n    material: lamp\n"
             "  - type: mesh\n    path: teapot.obj\n    material: g\n"
             "  - type: mesh\n    path: quad.obj\n    material: g\n"
             "  - type: mesh\n    path: teapot.obj\n    material: g\n    transform:\n        translate: [2.2, 0, 0.5]\n"
             "  - type: xz_rect\n    x: [-6, 6]\n    z: [-6, 6]\n    k: -0.6\n    material: a\n"
             "  - type: sphere\n    center: [-1.6, 0.2, 0.8]\n    radius: 0.5\n    material: g\n")
    name = "stale_%d.yaml" % int(enclosed)
    (tmp_path / name).write_text(
        "film:\n    width: 48\n    height: 48\n    samples: 4\n    output: o.png\n"
        "camera:\n    position: [0.5, 1.5, 8]\n    look_at: [0, 0.4, 0]\n    up: [0, 1, 0]\n    fov: 40\n    aperture: 0\n    focal_distance: 8\n"
        "    background: [0.6, 0.7, 0.9]\n"
        "materials:\n  - name: a\n    type: lambertian\n    albedo: [0.8, 0.4, 0.3]\n  - name: b\n    type: metal\n    albedo: [0.8, 0.8, 0.7]\n"
        "    roughness: 0.1\n  - name: g\n    type: dielectric\n    ior: 1.5\n  - name: lamp\n    type: diffuse_light\n    albedo: [1, 0.9, 0.8]\n"
        "    strength: 4\n  - name: sky\n    type: diffuse_light\n    albedo: [0.5, 0.6, 0.8]\n    strength: 1\n"
        "objects:\n" + objs)
    return str(tmp_path / name)
