"""The two forms of the attention kernels (dalm_amd/csrc/attn.hip: register-staged tiles + transposed LDS copies vs LDS-DMA
stages + transpose reads) restate the same arithmetic: on the same inputs every output - o, dq, dk, dv, padded and packed layouts,
head widths 128 / 64, causal + left padding, encoder masks, dropout, partial blocks, an empty sequence - is BIT-identical.
The form is chosen once per process (DALM_ATTN_FWD / DALM_ATTN_DKDV), hence two subprocesses of tools/attn_ab.py.
Reference call site of the operation: self.generator_model(...) / self.retriever_model(...), dalm/models/rag_e2e_base_model.py:84-106."""
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_second_forms_write_the_first_forms_bits(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    outs = []
    for tag, extra in (("first", {"DALM_ATTN_FWD": "1", "DALM_ATTN_DKDV": "1"}), ("second", {})):
        env = {k: v for k, v in os.environ.items() if k not in ("DALM_ATTN_FWD", "DALM_ATTN_DKDV")}
        env.update(extra)
        out = tmp_path / f"{tag}.pt"
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "attn_ab.py"), "--out", str(out)], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(torch.load(out))
    a, b = outs
    assert a.keys() == b.keys() and len(a) >= 68
    for k in a:
        assert torch.isfinite(a[k]).all(), k
        assert float(a[k].abs().max()) > 0, k
        assert torch.equal(a[k], b[k]), f"{k}: {(a[k] != b[k]).sum().item()} of {a[k].numel()} elements differ"
