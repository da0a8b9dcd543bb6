"""The whole-path parity cases shared by tests/golden/make_fixtures.py (which renders them with the reference's
own renderer), tests/test_oracle.py (oracle == reference, bit for bit) and tests/test_gpu_parity.py (HIP path vs
the same states)."""

# tests/03_volume (the reference's own assets): name -> (shader, resolution, samples, bounces)
CASES = {
    "vol_64_1": ("volpathtrace", 64, 1, 64),      # samples == 1: pixel-centre preview branch
    "vol_64_4": ("volpathtrace", 64, 4, 64),
    "vol_96_16": ("volpathtrace", 96, 16, 64),
    "path_64_4": ("pathtrace", 64, 4, 4),
    "naive_64_4": ("naive", 64, 4, 4),
    "eye_64_2": ("eyelight", 64, 2, 4),
    "normal_64_2": ("normal", 64, 2, 4),
    "texcoord_64_2": ("texcoord", 64, 2, 4),
    "color_64_2": ("color", 64, 2, 4),
}

# substitute scenes (tests/golden/make_scenes.py): name -> (scene, shader, resolution, samples, bounces, noimplicit_mis)
EXTRA = {
    # glossy + normal maps + subdivision surfaces (BASELINE config 1: tests/01_surface with one substituted cage)
    "surf_path_96_4": ("01_surface_min/surface_min.json", "pathtrace", 96, 4, 4, False),
    "surf_normal_96_1": ("01_surface_min/surface_min.json", "normal", 96, 2, 4, False),
    "surf_eye_96_2": ("01_surface_min/surface_min.json", "eyelight", 96, 2, 4, False),
    # the same scene through the volumetric shader: three Catmull-Clark cages (4 / 2 / 2 levels) and a displaced one
    "surf_subdiv_96_4": ("01_surface_min/surface_min.json", "volpathtrace", 96, 4, 8, False),
    # corner cases of tesselate_surfaces (non-manifold boundary, triangle cages, smooth = false, float / 8-bit displacement)
    "subdiv_path_96_4": ("08_subdiv_synth/subdiv_synth.json", "pathtrace", 96, 4, 4, False),
    "subdiv_normal_96_2": ("08_subdiv_synth/subdiv_synth.json", "normal", 96, 2, 4, False),
    # 144k-triangle mesh, two environments, rough subsurface refraction (config 3)
    "head_vol_96_4": ("05_head1ss_sub/head1ss_sub.json", "volpathtrace", 96, 4, 64, False),
    # voxel SDFs + box SDFs + SDF light (config 4)
    "sdf_implicit_96_4": ("06_gridsdf_synth/gridsdf_synth.json", "implicit", 96, 4, 4, False),
    "sdf_nomis_96_4": ("06_gridsdf_synth/gridsdf_synth.json", "implicit", 96, 4, 4, True),
    "sdf_normal_96_2": ("06_gridsdf_synth/gridsdf_synth.json", "implicit_normal", 96, 2, 4, False),
    # the same scene with the grids at config 4's size (96^3 + 64^3): the bench workload
    "sdf_full_implicit_96_4": ("06_gridsdf_full/gridsdf_full.json", "implicit", 96, 4, 4, False),
    # every sd_* primitive, reflective / transparent / gltfpbr / refractive lobes (rough and delta), opacity < 1
    "sdfn_implicit_128_8": ("07_sdfunction_synth/sdfunction_synth.json", "implicit", 128, 8, 6, False),
    "sdfn_nomis_128_4": ("07_sdfunction_synth/sdfunction_synth.json", "implicit", 128, 4, 6, True),
    "sdfn_normal_128_2": ("07_sdfunction_synth/sdfunction_synth.json", "implicit_normal", 128, 2, 4, False),
    # the same lobes on meshes, subsurface medium, an emissive mesh with a real BVH (100-hop light pdf walk)
    "lobes_path_96_8": ("03_volume_lobes/volume_lobes.json", "pathtrace", 96, 8, 8, False),
    "lobes_vol_96_8": ("03_volume_lobes/volume_lobes.json", "volpathtrace", 96, 8, 16, False),
    "lobes_naive_96_4": ("03_volume_lobes/volume_lobes.json", "naive", 96, 4, 8, False),
    "lobes_eye_96_2": ("03_volume_lobes/volume_lobes.json", "eyelight", 96, 2, 8, False),
}
