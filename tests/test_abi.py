"""C-ABI surface checks that need no GPU: the library loads, exports every symbol the header
declares, the ctypes table matches the header, and the product refuses CPU tensors loudly."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "dalm_hip.h"


def header_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(dalm_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from dalm_amd import _build, hip

    _build.build(verbose=False)  # hipcc cross-compiles for gfx950 without a GPU
    return hip.load()


def test_header_declares_expected_entry_points():
    names = header_functions()
    for required in ("dalm_marg_ce_fwd", "dalm_marg_ce_bwd", "dalm_sim_rowstats", "dalm_sim_grad",
                     "dalm_sim_matmul", "dalm_pool_l2norm_fwd", "dalm_pool_l2norm_bwd", "dalm_version",
                     "dalm_last_error_string"):
        assert required in names


def test_library_exports_every_header_symbol(lib):
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in include/dalm_hip.h but not exported"


def test_ctypes_table_matches_header(lib):
    from dalm_amd import hip

    assert sorted(hip.SIGNATURES) == header_functions()
    # argument counts: count commas in each prototype
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    for name, (_, args) in hip.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^)]*)\)" % name, text)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), f"{name}: header has {n} params, ctypes table {len(args)}"


def test_version_and_error_string(lib):
    assert lib.dalm_version() >= 100
    assert isinstance(lib.dalm_last_error_string(), bytes)


def test_argument_errors_are_reported_without_a_gpu(lib):
    # NULL pointers are rejected before any HIP call is made
    rc = lib.dalm_sim_matmul(None, None, 4, 4, 8, ctypes.c_float(1.0), None, 4, None)
    assert rc == -1
    assert b"null pointer" in lib.dalm_last_error_string()
    rc = lib.dalm_marg_ce_fwd(None, 0, 1, 2, 3, 6, 3, None, None, None, None, None, None, None)
    assert rc == -1
    # nf4: sizes of the two stores, and argument errors before any launch
    assert lib.dalm_nf4_packed_bytes(0) == 0 and lib.dalm_nf4_packed_bytes(64) == 32 and lib.dalm_nf4_packed_bytes(65) == 33
    assert lib.dalm_nf4_absmax_count(64) == 1 and lib.dalm_nf4_absmax_count(65) == 2
    assert lib.dalm_nf4_quantize(None, 0, 64, None, None, None) == -1
    assert lib.dalm_nf4_dequantize(None, None, -1, 0, None, None) == -2
    assert lib.dalm_nf4_quantize(None, 0, 0, None, None, None) == 0          # nothing to do is not an error


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the public functions raise on CPU tensors instead of computing elsewhere."""
    from dalm_amd.training.utils import train_utils as tu

    q = torch.randn(4, 8)
    with pytest.raises(RuntimeError, match="HIP|MI355X|GPU"):
        tu.get_cosine_sim(q, q, 100)
    with pytest.raises(RuntimeError, match="HIP|MI355X|GPU"):
        tu.get_nt_xent_loss(torch.randn(4, 4))
    from dalm_amd.models import nf4

    with pytest.raises(RuntimeError, match="HIP|MI355X|GPU"):
        nf4.quantize(torch.randn(128))
    with pytest.raises(RuntimeError, match="HIP|MI355X|GPU"):
        nf4.NF4Linear(torch.nn.Linear(64, 64))


def test_package_never_imports_oracle():
    """Product code must not import anything under oracle/."""
    for py in (ROOT / "dalm_amd").rglob("*.py"):
        src = py.read_text()
        assert "dalm_oracle" not in src and "import oracle" not in src and "from oracle" not in src, py


def test_header_is_plain_c(tmp_path):
    """include/dalm_hip.h must be consumable by a C compiler (cgo / JNI / ctypes-style bindings): compile a C
    translation unit that includes it and takes the address of every entry point."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    names = header_functions()
    src = tmp_path / "abi_check.c"
    body = "\n".join(f"  p[{i}] = (fn_t)&{n};" for i, n in enumerate(names))
    src.write_text(f'#include "dalm_hip.h"\ntypedef void (*fn_t)(void);\nfn_t p[{len(names)}];\n'
                   f'void fill(void) {{\n{body}\n}}\n')
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", str(ROOT / "include"), "-c", str(src),
                        "-o", str(tmp_path / "abi_check.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
