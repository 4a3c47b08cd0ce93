"""fic_encode_gray_{argb,u8}_multi: the one-call, one-thread, n_gpus-device entry the JNI host binds.
On a one-GPU box the shards are logical devices mapped onto the real one (FIC_FAKE_DEVICES, gather by device copies);
the RCCL side is exercised as far as one device allows (library load, communicator creation).  The result must not
depend on n_gpus."""
import os
import subprocess

import numpy as np
import pytest

import fic_amd
from fic_amd import capi, synth
from conftest import GOLDEN, same_f32

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def fake8():
    old = os.environ.get("FIC_FAKE_DEVICES")
    os.environ["FIC_FAKE_DEVICES"] = "8"
    yield
    if old is None:
        del os.environ["FIC_FAKE_DEVICES"]
    else:
        os.environ["FIC_FAKE_DEVICES"] = old
    capi.release_cache()


@pytest.mark.parametrize("size,B,n_iso", [(256, 8, 8), (256, 8, 1), (256, 4, 1), (128, 16, 8), (512, 8, 8)])
def test_multi_device_entry_equals_single_device(fake8, size, B, n_iso):
    g = synth.image_s(size, size, 31) if n_iso == 1 else synth.image_u(size, size, 32)
    one = capi.encode_gray_oneshot(g, B, None, n_iso)
    for n in (1, 2, 3, 8):
        got = capi.encode_gray_multi(g, B, None, n_iso, n)
        for k in ("idx_local", "iso", "qrows"):
            assert (got[k] == one[k]).all(), (n, k)
        assert same_f32(got["a"], one["a"]) and same_f32(got["b"], one["b"])


def test_multi_device_entry_windowed_and_against_oracle(fake8, oracle):
    g = np.load(os.path.join(GOLDEN, "lena_grey_256.npy"))
    for B, wK, n_iso in [(8, 2, 1), (8, 16, 1), (4, 8, 8)]:
        ref = oracle.encode_gray(oracle.gray_to_argb(g), 256, 256, B, wK, n_iso)
        got = capi.encode_gray_multi(g, B, wK, n_iso, 4)
        assert (got["idx_local"] == ref["info"][:, 0].astype(np.int32)).all() and (got["iso"] == ref["iso"]).all()
        assert same_f32(got["a"], ref["info"][:, 1]) and same_f32(got["b"], ref["info"][:, 2])
        assert (got["qrows"] == oracle.quantise_gray(ref["info"])).all()


def test_more_gpus_than_devices_is_refused_without_the_test_knob():
    os.environ.pop("FIC_FAKE_DEVICES", None)
    n = capi.lib().fic_device_count()
    with pytest.raises(fic_amd.FicError) as e:
        capi.encode_gray_multi(synth.image_u(128, 128, 1), 8, None, 1, n + 1)
    assert e.value.code == -4
    with pytest.raises(fic_amd.FicError):
        capi.encode_gray_multi(synth.image_u(128, 128, 1), 8, None, 1, 0)


def test_rccl_loads_and_creates_communicators():
    """librccl is loaded on first use; on this box: one communicator on device 0 (n >= 2 also runs the gather pattern)."""
    n = min(capi.lib().fic_device_count(), 8)
    capi.check(capi.lib().fic_debug_rccl_selftest(n))
    capi.release_cache()


def test_rccl_gather_error_path_closes_the_group_and_drops_the_communicators():
    """an error inside ncclGroupStart .. ncclGroupEnd must not leave the thread's group open or stale
    communicators cached.  The library's own gather function is driven with a send RCCL must refuse; afterwards the same
    pattern has to work on fresh communicators."""
    n = min(capi.lib().fic_device_count(), 8)
    capi.check(capi.lib().fic_debug_rccl_selftest(-n))
    capi.check(capi.lib().fic_debug_rccl_selftest(n))
    capi.release_cache()


def test_multi_device_entry_with_argb_input_and_peer_copy_gather(fake8, oracle):
    """ARGB input through the pinned staging copy (R channel, FC:596) and asynchronous per-device uploads; FIC_GATHER=copy."""
    g = synth.image_s(256, 256, 77)
    argb = oracle.gray_to_argb(g)
    one = capi.encode_gray_oneshot(g, 8, None, 8)
    old = os.environ.get("FIC_GATHER")
    os.environ["FIC_GATHER"] = "copy"
    try:
        for n in (2, 5):
            got = capi.encode_gray_multi(argb, 8, None, 8, n, width=256, height=256)
            for k in ("idx_local", "iso", "qrows"):
                assert (got[k] == one[k]).all(), (n, k)
            assert same_f32(got["a"], one["a"]) and same_f32(got["b"], one["b"])
    finally:
        if old is None:
            del os.environ["FIC_GATHER"]
        else:
            os.environ["FIC_GATHER"] = old
    assert capi.lib().fic_debug_gather_fallbacks() == 0


def test_cpp_driver_multi_device(tmp_path, oracle):
    """The compiled C++ driver (no Python, no torch in the process): n_gpus = 1 and k logical devices give the same .run bytes."""
    exe = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")
    if not os.path.exists(exe):
        pytest.fail("tests/cpp/host_mirror_test missing: run __graft_entry__.build()")
    g = synth.image_u(256, 256, 5)
    raw = tmp_path / "g.raw"
    raw.write_bytes(g.tobytes())
    outs = []
    for n in (1, 2, 5):
        out = tmp_path / f"m{n}.bin"
        subprocess.check_call([exe, "encode_multi", str(raw), "256", "256", "8", "61", "8", str(n), str(out)],
                              env={**os.environ, "FIC_FAKE_DEVICES": "8"})
        outs.append(out.read_bytes())
    assert outs[0] == outs[1] == outs[2]
    ref = oracle.encode_gray(oracle.gray_to_argb(g), 256, 256, 8, 61, 8, 0, 64)
    run = outs[0][:20 + 12 * 1024]
    assert run[20:20 + 12 * 64] == oracle.write_run_gray(ref["info"], 256, 256, 8, 61)[20:20 + 12 * 64]
