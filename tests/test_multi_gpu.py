"""vpt_multi (include/vpt.h): the multi-GPU fan-out inside libvpt_hip.so — one host thread per GPU, tile t -> devices[t % ndev],
frame assembly on devices[0].  A GPU box of this pool has ONE card, so the fan-out is exercised with one device and with
the same device listed several times (every part then has its own scene copy, stream, tile buffers and host thread, exactly
as on distinct cards; only the RCCL send / receive of the frame assembly is replaced by local copies, because RCCL refuses a
communicator with a duplicated device).  The results must equal the single-GPU entry point bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0], [0] * 8])
def test_multi_equals_single_gpu_bit_for_bit(vpt, scene03, dev03, devices):
    p = vpt.PathtraceParams(resolution=200, samples=6, shader="volpathtrace", bounces=64)   # 200 x 83: ragged tiles
    ref = scene03.make_state(p)
    dev03.pathtrace_samples(ref, p, 6)
    multi = vpt.MultiDeviceScene(scene03, devices)
    st = scene03.make_state(p)
    multi.pathtrace_samples(st, p, 2)      # progressive: 2 + 4 samples, state through the host arrays in between
    multi.pathtrace_samples(st, p, 8)      # capped at params.samples (yocto_pathtrace.cpp:1055)
    assert st.samples == 6 and (st.hits == 6).all()
    assert np.array_equal(st.image.view(np.uint32), ref.image.view(np.uint32)) and np.array_equal(st.rngs, ref.rngs)
    frame = multi.get_render(st.width, st.height)            # gathered + resolved on devices[0]
    assert np.array_equal(frame.view(np.uint32), vpt.get_render(ref).view(np.uint32))
    multi.pathtrace_samples(st, p, 1)                        # no-op once reached
    assert st.samples == 6


def test_multi_errors(vpt, scene03):
    with pytest.raises(vpt.VptError):
        vpt.MultiDeviceScene(scene03, [99])
    multi = vpt.MultiDeviceScene(scene03, [0])
    with pytest.raises(vpt.VptError):
        multi.get_render(64, 27)                             # nothing rendered yet
    import ctypes as C
    q = vpt.PathtraceParams(resolution=64, samples=4, shader="volpathtrace")
    fresh = scene03.make_state(q)
    with pytest.raises(vpt.VptError, match="no state"):
        multi.get_state(fresh)                               # nothing on the devices yet
    n = C.c_int(0)
    abi = q.to_abi()
    rc = vpt.hip.vpt_multi_render(multi.handle, C.byref(abi), 1, fresh.width, fresh.height, None, None, None, C.byref(n))
    assert rc == -1 and b"vpt_multi_set_state first" in vpt.hip.vpt_last_error()       # resident render without a resident state
    rc = vpt.hip.vpt_multi_render(multi.handle, C.byref(abi), 1, fresh.width, fresh.height, fresh.image.ctypes.data, None, None, C.byref(n))
    assert rc == -1                                          # host pointers: all three or none
    assert vpt.hip.vpt_multi_set_state(multi.handle, 0, 27, fresh.image.ctypes.data, fresh.hits.ctypes.data, fresh.rngs.ctypes.data, 0) == -1
    assert vpt.hip.vpt_multi_set_state(multi.handle, 64, 27, None, fresh.hits.ctypes.data, fresh.rngs.ctypes.data, 0) == -1
    multi.set_state(fresh)
    multi._resident = (fresh.width, fresh.height, 3)        # the caller claims a sample count the devices do not hold
    with pytest.raises(vpt.VptError):
        multi.render_resident(q, 1)
    assert multi.transport() == "local" and vpt.hip.vpt_multi_transport(None) == b""
    with pytest.raises(vpt.VptError):
        p = vpt.PathtraceParams(resolution=64, samples=2, shader="volpathtrace")
        st = scene03.make_state(p)
        abi = p.to_abi()
        abi.shader = 42
        import ctypes as C
        n = C.c_int(0)
        rc = vpt.hip.vpt_multi_render(multi.handle, C.byref(abi), 1, st.width, st.height, st.image.ctypes.data, st.hits.ctypes.data,
                                      st.rngs.ctypes.data, C.byref(n))
        assert rc == -4
        raise vpt.VptError(vpt.hip.vpt_last_error().decode())


def test_state_stays_resident_between_calls(vpt, scene03, dev03):
    """the tile state lives on the devices across sample batches.  Upload once, render in batches with no
    transfers, download on demand: the same bits as the single-GPU entry point; the host-pointer form skips its upload when
    the caller's arrays still are what it stored there, and sees every way a caller can hand it another state."""
    p = vpt.PathtraceParams(resolution=200, samples=12, shader="volpathtrace", bounces=64)
    ref = scene03.make_state(p)
    dev03.pathtrace_samples(ref, p, 12)
    multi = vpt.MultiDeviceScene(scene03, [0, 0, 0])
    st = scene03.make_state(p)
    multi.set_state(st)
    assert multi.render_resident(p, 5) == 5 and multi.render_resident(p, 4) == 9 and multi.render_resident(p, 100) == 12
    assert (st.image == 0).all() and st.samples == 0              # nothing came back on its own
    frame = multi.get_render(st.width, st.height)
    assert np.array_equal(frame.view(np.uint32), vpt.get_render(ref).view(np.uint32))
    multi.get_state(st)
    assert st.samples == 12 and (st.hits == 12).all()
    assert np.array_equal(st.image.view(np.uint32), ref.image.view(np.uint32)) and np.array_equal(st.rngs, ref.rngs)

    # host-pointer calls: upload skipped on the second call (same arrays, same checksum) - and the result is the same either way
    half = scene03.make_state(p)
    dev03.pathtrace_samples(half, p, 3)
    a = scene03.make_state(p)
    multi.pathtrace_samples(a, p, 1)
    assert multi.uploaded_parts() == 3
    multi.pathtrace_samples(a, p, 2)
    assert multi.uploaded_parts() == 0
    assert np.array_equal(a.image.view(np.uint32), half.image.view(np.uint32)) and np.array_equal(a.rngs, half.rngs)
    # the caller swaps in another state with the same sample count: a fresh state advanced by another handle
    b = scene03.make_state(p)
    dev03.pathtrace_samples(b, p, 3)
    b.image[:] = 0                                               # ... and edited
    multi.pathtrace_samples(b, p, 1)
    assert multi.uploaded_parts() == 3
    expect = scene03.make_state(p)
    dev03.pathtrace_samples(expect, p, 3)
    expect.image[:] = 0
    dev03.pathtrace_samples(expect, p, 1)
    assert np.array_equal(b.image.view(np.uint32), expect.image.view(np.uint32)) and np.array_equal(b.rngs, expect.rngs)
    # a reset (make_state: samples == 0) is always uploaded
    c = scene03.make_state(p)
    multi.pathtrace_samples(c, p, 12)
    assert np.array_equal(c.image.view(np.uint32), ref.image.view(np.uint32))
    # rendering on the devices without a matching resident state is an error, not a stale image
    with pytest.raises(vpt.VptError):
        multi._resident = (st.width, st.height, 5)
        multi.render_resident(p, 1)


def test_the_callers_arrays_are_the_state(vpt, scene03, dev03):
    """The reference reads the live pathtrace_state on every call (yocto_pathtrace.cpp:1081-1090).  The host-pointer form of
    vpt_multi_render may skip a device's upload only when that device provably holds the caller's data (include/vpt.h, the
    RULE): (1) an in-place edit of ONE pixel - of the radiance sum only, RNG state and hit count untouched, the kind of
    edit round 3's 64-pixel probe could not see - reaches the result, and only the device that owns the pixel uploads;
    (2) two state objects rendered alternately at equal sample counts through one handle (two shaders on one scene is how
    the drop-in meets this) never continue from each other's device buffers, although they agree on every background pixel."""
    p = vpt.PathtraceParams(resolution=200, samples=8, shader="volpathtrace", bounces=64)
    multi = vpt.MultiDeviceScene(scene03, [0, 0, 0])
    a = scene03.make_state(p)
    multi.pathtrace_samples(a, p, 2)
    ref = a.copy()
    # (1) one pixel, one word; its tile decides which device must upload (tile t -> devices[t % 3])
    y, x = 41, 77
    a.image[y, x, 1] += 0.25
    ref.image[y, x, 1] += 0.25
    multi.pathtrace_samples(a, p, 2)
    assert multi.uploaded_parts() == 1
    dev03.pathtrace_samples(ref, p, 2)
    assert np.array_equal(a.image.view(np.uint32), ref.image.view(np.uint32)) and np.array_equal(a.rngs, ref.rngs)
    multi.pathtrace_samples(a, p, 1)                             # untouched since: nothing to upload
    assert multi.uploaded_parts() == 0
    a.rngs[3, 5, 0] ^= 1                                         # an RNG word of another pixel
    multi.pathtrace_samples(a, p, 1)
    assert multi.uploaded_parts() == 1
    # (2) alternating states, same size and sample count at every call
    q = vpt.PathtraceParams(resolution=200, samples=8, shader="pathtrace", bounces=8)
    s1, s2 = scene03.make_state(p), scene03.make_state(q)
    r1, r2 = s1.copy(), s2.copy()
    for _ in range(3):
        multi.pathtrace_samples(s1, p, 2)
        assert multi.uploaded_parts() == 3
        multi.pathtrace_samples(s2, q, 2)
        assert multi.uploaded_parts() == 3
        dev03.pathtrace_samples(r1, p, 2)
        dev03.pathtrace_samples(r2, q, 2)
    assert s1.samples == s2.samples == 6
    assert np.array_equal(s1.image.view(np.uint32), r1.image.view(np.uint32)) and np.array_equal(s1.rngs, r1.rngs)
    assert np.array_equal(s2.image.view(np.uint32), r2.image.view(np.uint32)) and np.array_equal(s2.rngs, r2.rngs)
    # a copy at another address with identical contents is uploaded too (the addresses are part of the key) - same bits either way
    s3 = s2.copy()
    multi.pathtrace_samples(s3, q, 1)
    assert multi.uploaded_parts() == 3
    dev03.pathtrace_samples(r2, q, 1)
    assert np.array_equal(s3.image.view(np.uint32), r2.image.view(np.uint32))


def test_rccl_entry_points_run_on_one_gpu(vpt, scene03, dev03, monkeypatch):
    """VPT_MULTI_FORCE_RCCL=1: a one-rank communicator (ncclCommInitAll), the tile buffer sent to itself with grouped
    ncclSend / ncclRecv, ncclCommDestroy - every RCCL call of vpt_multi.cpp, on the one GPU this box has.  Same frame."""
    monkeypatch.setenv("VPT_MULTI_FORCE_RCCL", "1")
    p = vpt.PathtraceParams(resolution=200, samples=4, shader="volpathtrace", bounces=64)
    ref = scene03.make_state(p)
    dev03.pathtrace_samples(ref, p, 4)
    multi = vpt.MultiDeviceScene(scene03, [0])
    assert multi.transport() == "rccl"
    st = scene03.make_state(p)
    multi.pathtrace_samples(st, p, 4)
    for _ in range(2):                                           # the communicator is reused
        frame = multi.get_render(st.width, st.height)
        assert np.array_equal(frame.view(np.uint32), vpt.get_render(ref).view(np.uint32))
    multi.close()


def test_failed_rccl_init_falls_back_to_peer_copies(vpt, scene03, dev03, monkeypatch, tmp_path):
    """ncclCommInitAll failing (no P2P, busy device ...) must neither crash nor fail the handle: the frame assembly
    falls back to hipMemcpyPeerAsync.  A stub library whose ncclCommInitAll returns an error stands in for RCCL."""
    import subprocess
    src = tmp_path / "stub.c"
    src.write_text('int ncclCommInitAll(void** c, int n, const int* d) { for (int i = 0; i < n; i++) c[i] = (void*)(long)(i + 1); return 2; }\n'
                   'int destroyed = 0;\nint ncclCommDestroy(void* c) { destroyed++; return 0; }\nint ncclGroupStart() { return 2; }\nint ncclGroupEnd() { return 2; }\n'
                   'int ncclSend() { return 2; }\nint ncclRecv() { return 2; }\nconst char* ncclGetErrorString(int r) { return "stub failure"; }\n')
    lib = tmp_path / "librccl_stub.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(lib)])
    monkeypatch.setenv("VPT_MULTI_FORCE_RCCL", "1")
    monkeypatch.setenv("VPT_MULTI_RCCL_LIB", str(lib))
    p = vpt.PathtraceParams(resolution=128, samples=2, shader="volpathtrace", bounces=64)
    ref = scene03.make_state(p)
    dev03.pathtrace_samples(ref, p, 2)
    multi = vpt.MultiDeviceScene(scene03, [0])
    assert multi.transport() == "local"
    st = scene03.make_state(p)
    multi.pathtrace_samples(st, p, 2)
    assert np.array_equal(multi.get_render(st.width, st.height).view(np.uint32), vpt.get_render(ref).view(np.uint32))
    import ctypes as C
    assert C.c_int.in_dll(C.CDLL(str(lib)), "destroyed").value == 1   # the communicator the failed call had created was destroyed


def test_watchdog_fails_the_multi_render(vpt, monkeypatch):
    """A wave of the implicit kernel that gives up on its watchdog leaves an incomplete image: vpt_multi_render (what the
    host drop-in and ypathtrace call) must fail like vpt_render does.  VPT_K2_WATCHDOG_MS=0 makes every wave give up."""
    import os
    from conftest import GOLDEN
    scene = vpt.HostScene(os.path.join(GOLDEN, "scenes", "06_gridsdf_synth", "gridsdf_synth.json"))
    p = vpt.PathtraceParams(resolution=96, samples=4, shader="implicit", bounces=4)
    st = scene.make_state(p)
    good = vpt.MultiDeviceScene(scene, [0, 0])
    good.pathtrace_samples(st, p, 2)
    assert st.samples == 2
    monkeypatch.setenv("VPT_K2_WATCHDOG_MS", "0")
    bad = vpt.MultiDeviceScene(scene, [0, 0])
    st2 = scene.make_state(p)
    with pytest.raises(vpt.VptError, match="watchdog"):
        bad.pathtrace_samples(st2, p, 2)
    assert st2.samples == 0
    single = vpt.DeviceScene(scene, 0)
    with pytest.raises(vpt.VptError, match="watchdog"):
        single.pathtrace_samples(scene.make_state(p), p, 2)


@pytest.mark.parametrize("ndev", [2, 4, 8])
def test_distinct_gpus_when_the_box_has_them(vpt, scene03, dev03, ndev):
    """the RCCL gather between DISTINCT devices: runs the first time a multi-GPU node is available (skipped on the
    one-GPU boxes of this pool, where the tests above stand in for it)"""
    if vpt.device_count() < ndev:
        pytest.skip(f"needs {ndev} GPUs, this box has {vpt.device_count()}")
    p = vpt.PathtraceParams(resolution=320, samples=4, shader="volpathtrace", bounces=64)
    ref = scene03.make_state(p)
    dev03.pathtrace_samples(ref, p, 4)
    multi = vpt.MultiDeviceScene(scene03, list(range(ndev)))
    st = scene03.make_state(p)
    multi.pathtrace_samples(st, p, 4)
    assert np.array_equal(st.image.view(np.uint32), ref.image.view(np.uint32)) and np.array_equal(st.rngs, ref.rngs)
    assert np.array_equal(multi.get_render(st.width, st.height).view(np.uint32), vpt.get_render(ref).view(np.uint32))
    assert multi.transport() in ("rccl", "peer-copy")
