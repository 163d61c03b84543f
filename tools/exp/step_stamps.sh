# in-model stamps of the two attention bodies: cycles + clock inside the full denoising step (tools/step_ab.py under a -DTCX_ATTN_STAMP build)
. "$(dirname "${BASH_SOURCE[0]}")/with_experiments.sh" || exit 1     # patched scratch copy: the product sources carry no experiment switches
R=$GRAFT_REPO_ROOT
cd $R/trajectorycrafter_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -w -I. -DTCX_ATTN_STAMP -x hip -c attn_fwd.hip -o /tmp/attn_clk.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libtcx_aclk.so tcx_api.o /tmp/attn_clk.o norm.o elementwise.o conv.o conv_mfma.o groupnorm.o warp.o gemm.o || exit 1
cd $R
TCX_LIB=/tmp/libtcx_aclk.so python3 tools/step_ab.py 3 3 > gpurun_out/step_stamps.log 2>&1
grep -v ASTAMP gpurun_out/step_stamps.log | tail -12
python3 - <<'PY'
import re, statistics
by = {}
for l in open("gpurun_out/step_stamps.log"):
    m = re.match(r"ASTAMP body (\d+) D (\d+) wg \d+ wave (\d+) tiles (\d+) cycles (\d+) real (\d+)", l)
    if m and int(m.group(4)) > 200:
        by.setdefault((m.group(1), m.group(2)), []).append((int(m.group(5)), int(m.group(6))))
for k, v in sorted(by.items()):
    print("in-model body", k[0], "D", k[1], "stamps", len(v), "median cycles", statistics.median(c for c, _ in v), "median us", statistics.median(r for _, r in v) / 100,
          "clock GHz", round(statistics.median(c / r * 0.1 for c, r in v), 3))
PY
