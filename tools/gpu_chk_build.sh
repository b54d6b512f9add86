set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
AWPU_NO_BUILD=1 timeout -k 10 200 python bench.py --cpu-seconds 0 --no-extras > gpurun_out/chk_pre_$rep.json 2>/dev/null
python -c "
import json;d=json.loads([l for l in open('gpurun_out/chk_pre_$rep.json') if l.startswith('{')][0]);print('prebuilt', d['value'], d['roofline']['kernel_ms'])"
done
cp beamforming-lk_amd/libawpu_hip.so gpurun_out/keep_prebuilt.so
python3 -c "import __graft_entry__ as g; g.build()" > gpurun_out/chk_build.log 2>&1
cmp beamforming-lk_amd/libawpu_hip.so gpurun_out/keep_prebuilt.so && echo "identical binaries" || echo "binaries differ"
ls -la beamforming-lk_amd/libawpu_hip.so gpurun_out/keep_prebuilt.so
for rep in 1 2; do
timeout -k 10 200 python bench.py --cpu-seconds 0 --no-extras > gpurun_out/chk_re_$rep.json 2>/dev/null
python -c "
import json;d=json.loads([l for l in open('gpurun_out/chk_re_$rep.json') if l.startswith('{')][0]);print('rebuilt', d['value'], d['roofline']['kernel_ms'])"
done
timeout -k 10 300 python bench.py --workload headline > gpurun_out/chk_full.json 2>/dev/null
python -c "
import json;d=json.loads([l for l in open('gpurun_out/chk_full.json') if l.startswith('{')][0]);print('rebuilt full', d['value'], d['roofline']['kernel_ms'])"
rm -f gpurun_out/keep_prebuilt.so
