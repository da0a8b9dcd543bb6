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
