"""crnn_forward_device_noalloc (the entry point that allocates nothing: fixed workspace in the code object) against
crnn_forward_device, through the C ABI: identical raw arg-max, labels and lengths, on two calls in a row, the second
with a smaller input (the shared array keeps the first call's contents); plus its limits and error behaviour."""
import ctypes as C

import numpy as np
import pytest

from ocr_amd.crnn import _capi, keys

pytestmark = pytest.mark.gpu


def _run(lib, h, fn, lines, widths=None):
    import torch
    dev = torch.device("cuda", 0)
    n, _, w = lines.shape
    T = w // 4 + 1
    d = torch.from_numpy(lines).to(dev)
    raw = torch.full((n, T), -7, dtype=torch.int32, device=dev)
    lab = torch.full((n, T), -7, dtype=torch.int32, device=dev)
    ln = torch.full((n,), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()          # inputs and outputs are complete before the library's own stream touches them
    wv = None if widths is None else np.ascontiguousarray(widths, np.int32)
    free0 = torch.cuda.mem_get_info(0)[0]
    rc = fn(h, d.data_ptr(), n, w, None if wv is None else wv.ctypes.data, raw.data_ptr(), lab.data_ptr(), ln.data_ptr(), 1)
    _run.freed_delta = free0 - torch.cuda.mem_get_info(0)[0]     # device bytes the call took (0 for an entry point that allocates nothing)
    assert rc >= 0, lib.crnn_last_error(h)
    return raw.cpu().numpy(), lab.cpu().numpy(), ln.cpu().numpy()


def test_noalloc_entry_point_equals_forward_device(synth_zh, synth_en):
    from ocr_amd.crnn.network_hip import CRNN
    lib = _capi.lib()
    rs = np.random.RandomState(77)
    for nclass, alpha, sd in ((5530, keys.alphabetChinese, synth_zh), (65, keys.alphabetEnglish, synth_en)):
        m = CRNN(32, 1, nclass, 256, alphabet=alpha, device=0, lanes=2)
        m.load_state_dict(sd)
        h = m._h
        d8 = np.zeros((1, 32, 64), np.uint8)
        if nclass == 5530:
            # before any device state exists the entry point refuses (it may not create it) ...
            rc = lib.crnn_forward_device_noalloc(h, 1 << 20, 1, 64, None, None, 1 << 20, 1 << 20, 1)
            assert rc == _capi.CRNN_ERR_STATE and b"crnn_stream" in lib.crnn_last_error(h)
        assert m.stream_ptr(0)            # ... and this is the set-up call that is allowed to allocate
        # two calls in a row on the shared workspace, the second smaller; then a ragged one
        big = rs.randint(0, 256, size=(3, 32, 280)).astype(np.uint8)
        small = rs.randint(0, 256, size=(1, 32, 96)).astype(np.uint8)
        rag = rs.randint(0, 256, size=(4, 32, 200)).astype(np.uint8)
        got_big = _run(lib, h, lib.crnn_forward_device_noalloc, big)
        first_delta = _run.freed_delta      # set-up (crnn_stream) has made every "first launch" already: see ensure_device
        assert first_delta == 0, "the first crnn_forward_device_noalloc call of a handle took %d bytes of device memory" % first_delta
        got_small = _run(lib, h, lib.crnn_forward_device_noalloc, small)
        assert _run.freed_delta == 0, "crnn_forward_device_noalloc took device memory"
        got_rag = _run(lib, h, lib.crnn_forward_device_noalloc, rag, widths=[200, 8, 131, 64])
        assert _run.freed_delta == 0, "crnn_forward_device_noalloc took device memory"
        for got, (lines, wd) in ((got_big, (big, None)), (got_small, (small, None)), (got_rag, (rag, [200, 8, 131, 64]))):
            want = _run(lib, h, lib.crnn_forward_device, lines, widths=wd)
            T = lines.shape[2] // 4 + 1
            for i in range(lines.shape[0]):
                Ti = T if wd is None else wd[i] // 4 + 1
                assert np.array_equal(got[0][i, :Ti], want[0][i, :Ti]), (nclass, lines.shape, i)
                assert got[2][i] == want[2][i] and np.array_equal(got[1][i, :got[2][i]], want[1][i, :want[2][i]])
            assert (got[2] >= 0).all() and (got[0][:, 0] >= 0).all()         # really written
        # the same small input again after the big one gives the same answer (nothing left over matters)
        again = _run(lib, h, lib.crnn_forward_device_noalloc, small)
        assert _run.freed_delta == 0
        assert all(np.array_equal(a, b) for a, b in zip(again, got_small))
        # ... while crnn_forward_device itself does allocate when it meets a shape it has no workspace for yet (the property
        # the no-alloc entry point exists to avoid): the measurement above can see it
        fresh = rs.randint(0, 256, size=(5, 32, 312)).astype(np.uint8)
        _run(lib, h, lib.crnn_forward_device, fresh)
        grew = _run.freed_delta
        assert grew > 0, "the allocation check is blind: crnn_forward_device grew a workspace and no device memory went"
        # beyond the fixed workspace: refused with the sizes in the message, nothing launched
        need = lib.crnn_noalloc_workspace_bytes(h, 64, 280)
        assert need > 64 << 20
        rc = lib.crnn_forward_device_noalloc(h, 1 << 20, 64, 280, None, None, 1 << 20, 1 << 20, 1)
        assert rc == _capi.CRNN_ERR_ARG and b"MiB" in lib.crnn_last_error(h)
        assert lib.crnn_noalloc_workspace_bytes(h, 1, 280) <= 64 << 20
        rc = lib.crnn_forward_device_noalloc(h, None, 1, 280, None, None, None, None, 1)
        assert rc == _capi.CRNN_ERR_ARG
