"""CPU, build container only: the reference's weight hand-off formats end to end.  The reference module is imported by path,
given seeded weights, exported exactly as mlt-cnn-python/codes/model2torchScript.py:37-48 does (eval, torch.jit.trace with a
(1,2,128,128) input and (1,) poc / qp, .save) and saved as a `.pth` the way the training code does ({'params': ...} with
'module.' prefixes, model2torchScript.py:23-32).  tools/convert_weights.py must turn BOTH into the same MLTW blob the
in-memory route gives."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ARCH_DIR = "/root/reference/mlt-cnn-python/codes/models/archs"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_ARCH_DIR), reason="reference tree not mounted (GPU box)")


def _load(path, name):
    sys.dont_write_bytecode = True  # never drop .pyc files into the read-only reference tree
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("size", (128, 32))
def test_pt_and_pth_routes_give_the_same_blob(pkg, tmp_path, size):
    import torch
    arch = pkg.synth.arch_for_size(size)
    if arch == 0:
        model = _load(os.path.join(REF_ARCH_DIR, "mlt_ctu_or_pq_arch.py"), "ref_ctu_ts").GapBigMltCtuORPQ()
    else:
        model = _load(os.path.join(REF_ARCH_DIR, "mlt_cu_or_pq_arch.py"), "ref_cu_ts").GapBigMltCuORPQ()
    sd = pkg.synth.make_state_dict(arch, 77)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    model.eval()
    want = pkg.weights.pack_blob(arch, sd)
    # model2torchScript.py:37-48
    inp = torch.cat((torch.rand(1, 1, 128, 128), torch.rand(1, 1, 128, 128)), 1)
    traced = torch.jit.trace(model, (inp, torch.rand(1), torch.rand(1)))
    pt = tmp_path / f"MLTORPQ_splitMode_{size}.pt"
    traced.save(str(pt))
    pth = tmp_path / "net_1000.pth"
    torch.save({"params": {"module." + k: v for k, v in model.state_dict().items()}}, str(pth))
    for src, out in ((pt, tmp_path / "from_pt"), (pth, tmp_path / "from_pth")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), "--size", str(size), str(src), str(out)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        got = open(out / f"MLTORPQ_splitMode_{size}.mltw", "rb").read()
        assert got == want, f"{src.name}: blob differs from the in-memory route"
