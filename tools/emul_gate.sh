set -x
python tools/emul_gate.py > gpurun_out/r05_emul_gate.txt 2>&1; echo rc=$? >> gpurun_out/r05_emul_gate.txt
for v in 0 1 2 3; do
  echo "== variant $v" >> gpurun_out/r05_emul_units.txt
  JATTS_RESUNIT_EMUL_VARIANT=$v python tools/bench_unit.py --all --dtype emul >> gpurun_out/r05_emul_units.txt 2>&1
done
echo "== f32" >> gpurun_out/r05_emul_units.txt
python tools/bench_unit.py --all --dtype f32 >> gpurun_out/r05_emul_units.txt 2>&1
echo "== split" >> gpurun_out/r05_emul_units.txt
python tools/bench_unit.py --all --dtype split >> gpurun_out/r05_emul_units.txt 2>&1
tail -5 gpurun_out/r05_emul_gate.txt
