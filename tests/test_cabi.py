"""The C-ABI shared library loads on a CPU-only box and exports every symbol that
include/wfhip.h declares; the Python binding table covers the same set; numeric entry
points fail loudly (no CPU fallback) when no HIP device is present."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_functions():
    text = (ROOT / "include" / "wfhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wf_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    from waveforms_amd.csrc.build import build

    return build(verbose=False)


def test_header_symbols_exported(built_lib):
    names = declared_functions()
    assert len(names) >= 20
    handle = ctypes.CDLL(str(built_lib))
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, f"declared in wfhip.h but not exported: {missing}"


def test_binding_table_matches_header(built_lib):
    from waveforms_amd import _hip

    assert sorted(_hip.SIGNATURES) == declared_functions()
    lib = _hip.lib()
    assert lib.wf_version().decode().startswith("waveforms-amd")
    # pure host helpers of the ABI are callable without a GPU
    assert lib.wf_fir_out_len(3, 8, 65) == 65          # numpy "same": the longer operand
    assert lib.wf_fir_out_len(511, 8, 65) == 4096
    assert lib.wf_viterbi4_state_bytes(2) > 0 and lib.wf_viterbi4_state_bytes(65) == -1
    cfg = _hip.LinkConfig()
    cfg.nsym, cfg.sps, cfg.ntaps, cfg.mf_nfilt, cfg.timing_offset = 1000, 8, 65, 3, -1
    assert lib.wf_link_workspace_bytes(ctypes.byref(cfg)) > 1000 * 8 * 24


def test_geometry_entry_points_reject_out_of_range_sps(built_lib):
    """sps = 0 and sps = 600 used to divide by zero on the host (SIGFPE) inside the fused
    modulator's geometry; every exported way in now returns a status instead."""
    from waveforms_amd import _hip

    lib = _hip.lib()
    tl, spt, nt = ctypes.c_int64(7), ctypes.c_int64(7), ctypes.c_int64(7)
    for sps in (0, 1, 257, 600, -3):
        assert lib.wf_mod_tile_geometry(sps, 65, 100000, ctypes.byref(tl), ctypes.byref(spt), ctypes.byref(nt)) == _hip.WF_ERR_VALUE
        assert (tl.value, spt.value, nt.value) == (0, 0, 0)
        cfg = _hip.LinkConfig()
        cfg.nsym, cfg.sps, cfg.ntaps, cfg.mf_nfilt, cfg.mf_ntaps, cfg.timing_offset = 1 << 20, sps, 65, 3, 9, -1
        assert lib.wf_link_stream_workspace_bytes(ctypes.byref(cfg), 1 << 16) == -1
        info = (ctypes.c_int64 * 8)()
        assert lib.wf_link_stream_layout(ctypes.byref(cfg), 1 << 16, 0, info) == _hip.WF_ERR_VALUE
        assert lib.wf_link_stream_interior(ctypes.byref(cfg), 1 << 16, 1) == 0
    assert lib.wf_mod_tile_geometry(8, 0, 100000, None, None, None) == _hip.WF_ERR_VALUE
    assert lib.wf_mod_tile_geometry(8, 65, 0, None, None, None) == _hip.WF_ERR_VALUE
    assert lib.wf_mod_tile_geometry(8, 65, 100000, ctypes.byref(tl), ctypes.byref(spt), ctypes.byref(nt)) == 0
    assert (tl.value, spt.value) == (8192, 1024) and nt.value == 98


def test_link_config_layout_matches_c(tmp_path):
    """ctypes mirror of wf_link_config == what a C compiler lays out from the header."""
    import subprocess

    from waveforms_amd import _hip

    fields = [f[0] for f in _hip.LinkConfig._fields_]
    src = tmp_path / "layout.c"
    prints = "\n".join(f'    printf("%zu\\n", offsetof(wf_link_config, {f}));' for f in fields)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "wfhip.h"\nint main(void) {\n'
                   '    printf("%zu\\n", sizeof(wf_link_config));\n' + prints + "\n    return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", str(ROOT / "include"), "-o", str(exe), str(src)])
    nums = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert nums[0] == ctypes.sizeof(_hip.LinkConfig)
    assert nums[1:] == [getattr(_hip.LinkConfig, f).offset for f in fields]


def test_state_block_sizes_match_the_header():
    """The Python layer allocates the carried detector / stream state by the header's own constants."""
    from waveforms_amd import _hip

    text = (ROOT / "include" / "wfhip.h").read_text()
    for name in ("WF_CPM_STATE_BYTES", "WF_CPM_STREAM_STATE_BYTES"):
        assert int(re.search(rf"#define {name} (\d+)", text).group(1)) == getattr(_hip, name)


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from waveforms.cpm.modulate import cpm_modulate
    from waveforms.cpm.soqpsk import freq_pulse_soqpsk_tg
    from waveforms.glfsr import PNSequence
    from waveforms.viterbi.algorithm import SOQPSKTrellisDetector

    with pytest.raises(RuntimeError, match="HIP device"):
        cpm_modulate(np.zeros(4, dtype=np.int8), 0.25, freq_pulse_soqpsk_tg(8), 8)
    with pytest.raises(RuntimeError, match="HIP device"):
        PNSequence(9).generate_sequence()
    with pytest.raises(RuntimeError, match="HIP device"):
        SOQPSKTrellisDetector().iteration(np.zeros(3, dtype=np.complex128))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under waveforms_amd/ or waveforms/ may
    reference it."""
    offenders = []
    for pkg in ("waveforms_amd", "waveforms"):
        for path in (ROOT / pkg).rglob("*"):
            if path.suffix in (".py", ".hip", ".h", ".cpp") and re.search(r"\boracle\b", path.read_text()):
                offenders.append(str(path.relative_to(ROOT)))
    assert not offenders, offenders
