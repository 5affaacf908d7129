"""CPU: weight converter round trip and the C++ host mirror (builds with plain g++, links the C ABI)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blob_round_trip_and_checkpoint_conventions(pkg, tmp_path):
    sd = pkg.synth.make_state_dict(1, 3)
    blob = pkg.weights.pack_blob(1, sd)
    arch, back = pkg.weights.unpack_blob(blob)
    assert arch == 1
    for k, v in sd.items():
        if k.endswith("num_batches_tracked") or k.startswith("bn1."):
            continue
        assert np.array_equal(back[k], v), k
    # model2torchScript.py:23-32 conventions: {'params': ...} wrapper and 'module.' prefixes
    wrapped = {"params": {"module." + k: v for k, v in sd.items()}}
    assert pkg.weights.from_checkpoint(wrapped, 1) == blob


def test_convert_weights_cli(pkg, tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), "--size", "32", "--synthetic", "10", str(tmp_path)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    data = open(tmp_path / "MLTORPQ_splitMode_32.mltw", "rb").read()
    assert data == pkg.weights.synthetic_blob(1, 10)


def test_cpp_host_mirror_builds_and_links(pkg, tmp_path):
    lib = pkg.build.build_lib()
    exe = str(tmp_path / "callsite_demo")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", os.path.join(ROOT, "host", "callsite_demo.cpp"), "-o", exe,
           "-L" + os.path.dirname(lib), "-lmltcnn_hip", "-Wl,-rpath," + os.path.dirname(lib)]
    assert subprocess.run(cmd, capture_output=True, text=True).returncode == 0
    out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib"))
    assert out.returncode == 0 and "abi 4, logits(128) 9, logits(32) 15" in out.stdout


def test_gate_truth_table_matches_the_reference_condition(pkg, tmp_path):
    """SplitPredictor::gate against EncCu.cpp:746-756 (useCNN): luma / joint tree only, never on I slices, square CUs of an
    enabled size only (reference: 128; the 64 / 32 / 16 clauses are commented out at :754), CU entirely inside the picture."""
    lib = pkg.build.build_lib()
    src = tmp_path / "gate_table.cpp"
    src.write_text(r'''
#include "mlt_split_predictor.hpp"
int main(int argc, char **argv) {
  unsigned mask = (unsigned)std::strtoul(argv[1], nullptr, 0);
  mlt::SplitPredictor cnn("/nonexistent", 0, mask);   // no GPU / no weights here: gate() is pure host logic
  const int picW = 1920, picH = 1080;
  const int sizes[] = {128, 64, 32, 16, 8};
  for (int chType = 0; chType < 2; ++chType)
    for (int intra = 0; intra < 2; ++intra)
      for (int w : sizes)
        for (int h : sizes)
          for (int x : {0, 1792, 1856, 1900})
            for (int y : {0, 896, 960, 1024, 1072})
              std::printf("%d %d %d %d %d %d %d\n", chType, intra, w, h, x, y, (int)cnn.gate(chType, intra != 0, x, y, w, h, picW, picH));
  return 0;
}
''')
    exe = str(tmp_path / "gate_table")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "host"), str(src), "-o", exe,
           "-L" + os.path.dirname(lib), "-lmltcnn_hip", "-Wl,-rpath," + os.path.dirname(lib)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr

    def reference_use_cnn(chType, intra, cuw, cuh, cux, cuy, enabled, picW=1920, picH=1080):
        use = False
        if chType == 0 and not intra:                                   # EncCu.cpp:753
            if cuw == cuh and cuw in enabled:                           # :754 (128 only upstream)
                if cux + cuw <= picW and cuy + cuh <= picH:             # :755
                    use = True
        return use

    for mask, enabled in ((0x1, {128}), (0x0, {128}), (0xF, {128, 64, 32, 16}), (0x6, {64, 32})):
        out = subprocess.run([exe, str(mask)], capture_output=True, text=True, env=dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib"))
        assert out.returncode == 0
        rows = [tuple(map(int, l.split())) for l in out.stdout.splitlines() if l and l[0].isdigit()]
        assert len(rows) == 2 * 2 * 5 * 5 * 4 * 5
        positives = 0
        for chType, intra, w, h, x, y, got in rows:
            want = reference_use_cnn(chType, bool(intra), w, h, x, y, enabled)
            assert bool(got) == want, (mask, chType, intra, w, h, x, y)
            positives += want
        assert positives > 0


def test_traffic_json_names_are_the_profile_names_of_the_runtime():
    """scripts/make_traffic_json.py keys the PMC traffic by the launch names the runtime profiles under (bench.py looks `name@batch` up); a name the
    runtime truncates (mlt_profile_entry.name holds 47 characters) or renames silently turns `roofline.traffic` into null.  Every fused-launch name
    the script emits must be what mlt_api.cpp passes to prof_begin, cut to 47 characters."""
    import re
    src = open(os.path.join(ROOT, "fastintercu-vvc_amd", "csrc", "mlt_api.cpp")).read()
    runtime = {m[:47] for m in re.findall(r'prof_begin\("([^"]+)"', src)}
    script = open(os.path.join(ROOT, "scripts", "make_traffic_json.py")).read()
    emitted = set(re.findall(r'(?:name = |else )"([^"]+)"(\[:47\])?', script))
    fused = {(n[:47] if cut else n) for n, cut in emitted if n.startswith(("layer0_stream", "layer1_stream", "heads"))}
    assert len(fused) == 4
    assert fused and fused <= runtime, (fused - runtime, sorted(runtime)[:12])
