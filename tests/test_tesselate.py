"""tesselate_surfaces (libs/yocto_pathtrace/yocto_pathtrace.cpp:1119-1280): Catmull-Clark subdivision of
face-varying cages, split_facevarying, quads_to_triangles, displacement, smooth normals - and the OBJ reader that feeds it.

Pinning: tests/golden/substitute_stats.json holds the FNV-1a hashes the REFERENCE printed for positions / normals / texcoords /
triangles of every tesselated shape (oracle/_ref/ref_driver --stats, tests/golden/make_fixtures.py): the reference's own cages
of tests/01_surface (cube 4 levels, spot 2, suzanne 2) + a displaced sphere, and the hand-made corner cases of 08_subdiv_synth.
The host path must reproduce the hashes (CPU tests); the device path (vpt_subdivide_vertices) must reproduce the host path bit
for bit (GPU tests)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

SCENES = ["01_surface_min/surface_min.json", "08_subdiv_synth/subdiv_synth.json"]


def _stats(vpt, scene_file, **kw):
    return json.loads(vpt.HostScene(os.path.join(GOLDEN, "scenes", scene_file), **kw).stats())


@pytest.mark.parametrize("scene_file", SCENES)
def test_tesselated_shapes_hash_equal_to_the_reference(vpt, scene_file):
    golden = json.load(open(os.path.join(GOLDEN, "substitute_stats.json")))[scene_file]
    mine = _stats(vpt, scene_file)
    subdivided = [s["shape"] for s in json.load(open(os.path.join(GOLDEN, "scenes", scene_file)))["subdivs"]]
    assert len(subdivided) == 4
    for i in subdivided:
        a, b = golden["shapes"][i], mine["shapes"][i]
        assert b["quads"] == 0 and b["triangles"] > 0                      # quads_to_triangles
        for key in ("positions", "normals", "texcoords", "triangles", "pos_fnv", "nrm_fnv", "uv_fnv", "tri_fnv", "bvh_nodes_fnv", "bvh_prims_fnv"):
            assert a[key] == b[key], (scene_file, i, key)
    assert mine == golden                                                  # and everything downstream of it (BVHs, lights)


def test_the_references_cages_have_the_expected_sizes(vpt):
    shapes = _stats(vpt, SCENES[0])["shapes"]
    # cube-subdiv: 6 quads, 4 levels -> 6 * 4^4 quads -> 3072 triangles; spot / suzanne: 2 levels; the displaced sphere keeps its 6144 quads
    assert shapes[4]["triangles"] == 6 * 4 ** 4 * 2 and shapes[2]["triangles"] == 6144 * 2
    assert shapes[1]["triangles"] == 6368 and shapes[3]["triangles"] == 15752
    corner = _stats(vpt, SCENES[1])["shapes"]
    assert corner[2]["normals"] == 0 and corner[2]["texcoords"] == 0       # smooth = false, cage without texcoords
    assert corner[4]["normals"] == 0                                       # displaced with smooth = false: normals dropped after the displacement


def _load_shape_via_scene(vpt, tmp_path, obj_text, name="a.obj", subdiv=None):
    """load an OBJ through load_scene (the only public way in): returns the scene's stats or raises VptError"""
    (tmp_path / name).write_text(obj_text)
    scene = {"asset": {"version": "4.2"}, "cameras": [{"name": "c", "aspect": 1.0}], "materials": [{"name": "m", "type": "matte", "color": [0.5, 0.5, 0.5]}],
             "shapes": [{"name": "s", "uri": name}], "instances": [{"name": "i", "shape": 0, "material": 0}]}
    if subdiv is not None:
        scene["subdivs"] = [dict(subdiv, shape=0, uri=name)]
    (tmp_path / "scene.json").write_text(json.dumps(scene))
    return json.loads(vpt.HostScene(str(tmp_path / "scene.json")).stats())


def test_obj_reader_follows_the_references_rules(vpt, tmp_path):
    # vertices are unified per distinct (position, texcoord, normal) triple in order of first appearance; one 4-corner face
    # turns every face into a quad (triangles become z == w); negative indices count from the end; comments and unknown
    # statements are skipped (yocto_modelio.cpp:1952-2072, 2349-2356)
    text = ("# a comment\nmtllib nothing.mtl\no thing\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 2 0 0 # trailing comment\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
            "usemtl x\nf 1/1 2/2 3/3 4/4\nf 2/1 5/2 3/3\ns off\nf -4/1 -1/2 -3/4\n")
    st = _load_shape_via_scene(vpt, tmp_path, text)["shapes"][0]
    # distinct triples: 1/1 2/2 3/3 4/4 | 2/1 5/2 (3/3 seen) | (2/1 seen) 5/2 seen, 3/4 new -> 7 vertices
    assert (st["positions"], st["texcoords"], st["normals"], st["quads"], st["triangles"]) == (7, 7, 0, 3, 0)
    tri_only = _load_shape_via_scene(vpt, tmp_path, "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 2 0\nf 1 2 3 4 5\n")["shapes"][0]
    assert (tri_only["triangles"], tri_only["quads"], tri_only["positions"]) == (3, 0, 5)   # a pentagon is fanned
    for bad, why in (("v 0 0\nf 1 1 1\n", "parse error"), ("v 0 0 0\nf 1 2 3\n", "parse error"), ("v 0 0 0\nv 1 0 0\n", "empty shape"),
                     ("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2 3\n", "parse error")):
        with pytest.raises(vpt.VptError, match=why):
            _load_shape_via_scene(vpt, tmp_path, bad)
    with pytest.raises(vpt.VptError, match="a.stl: unknown format"):
        _load_shape_via_scene(vpt, tmp_path, "solid\n", name="a.stl")
    # point / line elements are outside the hot-path scope: loaded, then rejected when the scene is flattened
    with pytest.raises(vpt.VptError):
        _load_shape_via_scene(vpt, tmp_path, "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\nl 1 2\n")
    # ... and in a subdivision cage they are refused at load time: the reference's get_fvquads (yocto_modelio.cpp:2446) skips such an
    # element without advancing its vertex cursor, so it would read every later face from the wrong vertices (deliberate divergence)
    with pytest.raises(vpt.VptError, match="line / point elements"):
        _load_shape_via_scene(vpt, tmp_path, "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nl 1 2\nf 1 2 3 4\n", subdiv={"subdivisions": 1})
    # a literal longer than any fixed window is ONE number (round 3 copied 63 characters and read the tail as the next coordinate)
    long_one = "1." + "0" * 80 + "5"
    st = _load_shape_via_scene(vpt, tmp_path, f"v 0 0 0\nv {long_one} 0 0\nv 0 1 0\nf 1 2 3\n")["shapes"][0]
    assert (st["positions"], st["triangles"]) == (3, 1)
    with pytest.raises(vpt.VptError, match="parse error"):   # the same literal where only two coordinates follow "v": not three numbers
        _load_shape_via_scene(vpt, tmp_path, f"v 0 0 0\nv {long_one} 0\nv 0 1 0\nf 1 2 3\n")
    with pytest.raises(vpt.VptError, match="parse error"):   # an index that overflows the reader's window
        _load_shape_via_scene(vpt, tmp_path, "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 " + "3" * 40 + "\n")


def test_a_subdiv_replaces_its_shape_entirely(vpt, tmp_path):
    # face-varying cage: 8 positions, 24 texcoords; one level -> 6 * 4 quads, split per distinct (position, normal, texcoord)
    cube = open(os.path.join(GOLDEN, "scenes", "01_surface_min", "subdivs", "cube-subdiv.obj")).read()
    flat = _load_shape_via_scene(vpt, tmp_path, cube)["shapes"][0]
    assert (flat["positions"], flat["quads"]) == (24, 6)
    one = _load_shape_via_scene(vpt, tmp_path, cube, subdiv={"subdivisions": 1, "smooth": True})["shapes"][0]
    assert (one["quads"], one["triangles"]) == (0, 48) and one["normals"] == one["positions"] == one["texcoords"]
    zero = _load_shape_via_scene(vpt, tmp_path, cube, subdiv={"subdivisions": 0})["shapes"][0]
    assert (zero["positions"], zero["triangles"]) == (24, 12)   # no level: the cage itself, its own normals, triangulated


def test_catmullclark_levels_are_consistent(vpt):
    """structure of one level (counts, Euler characteristic, boundary handling) on meshes small enough to check by hand"""
    quads = np.array([[0, 1, 4, 3], [1, 2, 5, 4], [3, 4, 7, 6], [4, 5, 8, 7]], np.int32)           # an open 2 x 2 grid
    verts = np.array([[x, y, 0.25 * x * y] for y in range(3) for x in range(3)], np.float32)
    q1, v1 = vpt.catmullclark(quads, verts)
    assert len(q1) == 16 and len(v1) == 9 + 12 + 4
    # a corner of an open mesh is creased: the mean of the midpoints of its two half boundary edges
    assert np.array_equal(v1[[0, 2, 6, 8], :2], np.float32([[0.125, 0.125], [1.875, 0.125], [0.125, 1.875], [1.875, 1.875]]))
    ql, vl = vpt.catmullclark(quads, verts[:, :2].copy(), lock_boundary=True)
    assert np.array_equal(vl[[0, 2, 6, 8]], verts[[0, 2, 6, 8], :2])                               # locked boundary: (x + x) / 2 == x
    assert np.array_equal(q1, ql)
    tri = np.array([[0, 1, 2, 2]], np.int32)                                                       # one triangle -> three quads around its centroid
    qt, vt = vpt.catmullclark(tri, np.float32([[0, 0, 0], [3, 0, 0], [0, 3, 0]]))
    assert len(qt) == 3 and len(vt) == 3 + 3 + 1 and (qt[:, 2] == 6).all()
    with pytest.raises(vpt.VptError):
        vpt.catmullclark(np.array([[0, 1, 2, 9]], np.int32), verts)


def test_host_normals_and_displacement_stages(vpt):
    """the single-stage entry points (host form) on meshes small enough to check by hand; the whole of tesselate_surfaces is pinned to the
    reference by the hash tests above"""
    pos = np.float32([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [2, 0, 1]])
    n4 = vpt.vertex_normals(pos, np.int32([[0, 1, 2, 3]]))
    assert np.array_equal(n4[:4], np.float32([[0, 0, 1]] * 4)) and np.array_equal(n4[4], np.float32([0, 0, 0]))   # an unused vertex keeps the zero vector
    n3 = vpt.vertex_normals(pos, np.int32([[0, 1, 2], [1, 4, 2]]))
    assert np.array_equal(n3[0], np.float32([0, 0, 1])) and abs(np.linalg.norm(n3[1]) - 1) < 1e-6 and n3[1][0] < 0     # vertex 1 blends both faces
    tri_as_quad = vpt.vertex_normals(pos, np.int32([[0, 1, 2, 2]]))                                                    # z == w: three corners add, not four
    assert np.array_equal(tri_as_quad[:3], np.float32([[0, 0, 1]] * 3))
    with pytest.raises(vpt.VptError):
        vpt.vertex_normals(pos, np.int32([[0, 1, 9]]))
    tex = np.zeros((2, 2, 4), np.uint8)
    tex[..., :3] = 255                                                                                                 # white, 8-bit: mean 1 - 0.5
    out = vpt.displace_vertices(tex, False, 0.5, pos[:4], np.float32([[0, 0, 1]] * 4), np.float32([[0.1, 0.2]] * 4))
    assert np.allclose(out, pos[:4] + np.float32([0, 0, 0.25]), rtol=0, atol=1e-6)   # the four bilinear weights sum to 1 within an ulp
    texf = np.full((2, 2, 4), 0.5, np.float32)                                                                         # float texels: no offset
    out = vpt.displace_vertices(texf, True, 2.0, pos[:4], np.float32([[0, 0, 1]] * 4), np.float32([[0.6, 0.7]] * 4))
    assert np.allclose(out, pos[:4] + np.float32([0, 0, 1.0]), rtol=0, atol=1e-6)


def _random_mesh(rng, n):
    """an n x n grid with holes, some cells split into two triangles, two cells welded into a bow tie"""
    idx = lambda x, y: y * (n + 1) + x
    quads = []
    for y in range(n):
        for x in range(n):
            r = rng.random()
            if r < 0.15:
                continue
            a, b, c, d = idx(x, y), idx(x + 1, y), idx(x + 1, y + 1), idx(x, y + 1)
            if r < 0.35:
                quads += [[a, b, c, c], [a, c, d, d]]
            else:
                quads.append([a, b, c, d])
    verts = rng.normal(size=((n + 1) ** 2, 3)).astype(np.float32)
    return np.array(quads, np.int32), verts


@pytest.mark.gpu
def test_device_normals_and_displacement_equal_the_host_stages_bit_for_bit(vpt, dev03):
    """vpt_vertex_normals / vpt_displace_vertices (csrc/vpt_subdiv.hip) against the host loops that the reference's hashes pin: same float bits
    on random meshes with holes, triangles as z == w quads, a welded bow tie, shared and unused vertices; 8-bit sRGB, 8-bit linear and float
    displacement textures, texture coordinates outside [0, 1) and on texel borders"""
    rng = np.random.default_rng(11)
    for n in (1, 3, 17, 60):
        quads, verts = _random_mesh(rng, n)
        if len(quads) == 0:
            continue
        verts = np.vstack([verts, rng.normal(size=(3, 3)).astype(np.float32)])                     # three vertices no face uses
        hq, dq = vpt.vertex_normals(verts, quads), vpt.vertex_normals(verts, quads, device=0)
        assert np.array_equal(hq.view(np.uint32), dq.view(np.uint32)), n
        tris = np.array([[q[0], q[1], q[3]] for q in quads] + [[q[2], q[3], q[1]] for q in quads if q[2] != q[3]], np.int32)
        ht, dt = vpt.vertex_normals(verts, tris), vpt.vertex_normals(verts, tris, device=0)
        assert np.array_equal(ht.view(np.uint32), dt.view(np.uint32)), n
        uv = rng.uniform(-1.5, 2.5, size=(len(verts), 2)).astype(np.float32)
        uv[::7] = np.round(uv[::7] * 8) / 8                                                         # exactly on texel borders of the 8 x 4 textures
        for texels, linear in ((rng.integers(0, 256, size=(4, 8, 4), dtype=np.uint8), False), (rng.integers(0, 256, size=(4, 8, 4), dtype=np.uint8), True),
                               (rng.uniform(0, 2, size=(5, 3, 4)).astype(np.float32), True)):
            hd = vpt.displace_vertices(texels, linear, 0.37, verts, ht, uv)
            dd = vpt.displace_vertices(texels, linear, 0.37, verts, ht, uv, device=0)
            assert np.array_equal(hd.view(np.uint32), dd.view(np.uint32)), (n, texels.dtype, linear)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,lock", [(3, False), (2, True), (3, True), (2, False)])
def test_device_levels_equal_host_levels_bit_for_bit(vpt, dev03, dim, lock):
    rng = np.random.default_rng(5 + dim + 2 * lock)
    for n in (1, 2, 7, 40):
        quads, verts = _random_mesh(rng, n)
        if len(quads) == 0:
            continue
        verts = np.ascontiguousarray(verts[:, :dim])
        for level in range(3):
            qh, vh = vpt.catmullclark(quads, verts, lock)
            qd, vd = vpt.catmullclark(quads, verts, lock, device=0)
            assert np.array_equal(qh, qd) and np.array_equal(vh.view(np.uint32), vd.view(np.uint32)), (n, level)
            quads, verts = qh, vh
    cube = np.array([[0, 1, 2, 3], [4, 5, 6, 7], [1, 4, 7, 2], [5, 0, 3, 6], [3, 2, 7, 6], [1, 0, 5, 4]], np.int32)
    pos = np.float32([[-1, 0, 1], [1, 0, 1], [1, 2, 1], [-1, 2, 1], [1, 0, -1], [-1, 0, -1], [-1, 2, -1], [1, 2, -1]])[:, :dim].copy()
    for level in range(6):                                                                          # 6 * 4^6 = 24 576 quads
        qh, vh = vpt.catmullclark(cube, pos, lock)
        qd, vd = vpt.catmullclark(cube, pos, lock, device=0)
        assert np.array_equal(vh.view(np.uint32), vd.view(np.uint32))
        cube, pos = qh, vh


@pytest.mark.gpu
@pytest.mark.parametrize("scene_file", SCENES)
def test_device_tesselated_scenes_equal_the_reference(vpt, dev03, scene_file):
    golden = json.load(open(os.path.join(GOLDEN, "substitute_stats.json")))[scene_file]
    assert _stats(vpt, scene_file, tess_device=0) == golden


def _random_cage_obj(rng, n, with_uv, holes=True):
    """an n x n grid cage with holes, some cells as two triangles, some as pentagon + triangle fans; face-varying texcoords"""
    lines, verts = [], {}
    for y in range(n + 1):
        for x in range(n + 1):
            verts[(x, y)] = len(verts) + 1
            lines.append(f"v {x * 0.02 + rng.normal() * 0.002:.6f} {0.03 + rng.normal() * 0.01:.6f} {y * 0.02 + rng.normal() * 0.002:.6f}")
    nvt = 0
    faces = []
    vt_of = {}   # a vertex's shared texcoord: corners of one face (distinct vertices) never share an index, which the reference requires
                 # (a quad whose last two texcoord indices coincide would count as a triangle in one topology and not in the other)
    for y in range(n):
        for x in range(n):
            r = rng.random()
            if holes and r < 0.15:
                continue
            a, b, c, d = verts[(x, y)], verts[(x + 1, y)], verts[(x + 1, y + 1)], verts[(x, y + 1)]
            cells = [[a, b, c], [a, c, d]] if r < 0.35 else [[a, b, c, d]]
            for cell in cells:
                if with_uv:
                    idx = []
                    for v in cell:
                        if v not in vt_of or rng.random() < 0.3:   # a seam: this corner gets a texcoord of its own
                            nvt += 1
                            lines.append(f"vt {rng.random():.5f} {rng.random():.5f}")
                            if v not in vt_of:
                                vt_of[v] = nvt
                            idx.append(nvt)
                        else:
                            idx.append(vt_of[v])
                    faces.append("f " + " ".join(f"{v}/{t}" for v, t in zip(cell, idx)))
                else:
                    faces.append("f " + " ".join(str(v) for v in cell))
    return "\n".join(lines + faces) + "\n", len(faces)


@pytest.mark.parametrize("seed", range(12))
def test_random_cages_tesselate_like_the_reference_does_now(vpt, tmp_path, seed):
    """Live comparison where the reference is present (the build container): random cages - open boundaries, holes, triangles,
    face-varying texture charts, 1 to 3 levels, smooth on / off, displaced or not - through the reference's load_scene +
    tesselate_surfaces (oracle/_ref/ref_driver --stats) and through ours: the same hashes of all four arrays."""
    import oracle_lib as O
    if not O.have_reference():
        pytest.skip("oracle/_ref is not built here (the committed fixtures pin the same path)")
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(1, 7))
    with_uv = bool(seed % 2 == 0)
    text, nfaces = _random_cage_obj(rng, n, with_uv)
    if nfaces == 0:
        text, nfaces = _random_cage_obj(rng, n, with_uv, holes=False)
    (tmp_path / "cage.obj").write_text(text)
    tex = os.path.join(GOLDEN, "scenes", "shared_textures", "bumps-displacement.png")
    subdiv = {"name": "s", "shape": 0, "uri": "cage.obj", "subdivisions": int(rng.integers(1, 4)), "smooth": bool(rng.random() < 0.6)}
    if with_uv and rng.random() < 0.7:
        subdiv.update(displacement=0.01, displacement_tex=0)
    scene = {"asset": {"version": "4.2"}, "cameras": [{"name": "c", "aspect": 1.0, "frame": [1, 0, 0, 0, 1, 0, 0, 0, 1, 0.05, 0.05, 0.4]}],
             "textures": [{"name": "t", "uri": tex}], "materials": [{"name": "m", "type": "matte", "color": [0.5, 0.5, 0.5]}],
             "shapes": [{"name": "s", "uri": "cage.obj"}], "subdivs": [subdiv], "instances": [{"name": "i", "shape": 0, "material": 0}],
             "environments": [{"name": "e", "emission": [1, 1, 1]}]}
    path = tmp_path / "scene.json"
    path.write_text(json.dumps(scene))
    *_, ref = O.reference_render(str(path), "eyelight", 16, 1, 4, stats=True, workdir=str(tmp_path))
    mine = json.loads(vpt.HostScene(str(path)).stats())
    assert mine["shapes"] == ref["shapes"], (seed, subdiv, mine["shapes"], ref["shapes"])
    assert mine["scene_bvh"] == ref["scene_bvh"]
