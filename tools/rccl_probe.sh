# usage (on a GPU box): bash tools/rccl_probe.sh -- runs the one-rank RCCL window of tests/cpp/test_shim.cc four times, alternating the
# ROCm 7.2 librccl (the library rpath) and the RCCL that PyTorch bundles, with a 25-s kill: on part of the MI355X pool ncclCommInitRank of the
# 7.2 build never returns.
cd $GRAFT_REPO_ROOT
g++ -std=c++17 -O1 tests/cpp/test_shim.cc -o /tmp/test_shim -Lgyeeta_amd/lib -lgysketch -Wl,-rpath,$PWD/gyeeta_amd/lib -pthread -Wl,-rpath,/opt/rocm/lib
TL=$(python3 -c "import os,torch; print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
ls /sys/class/net; rocm-smi --showid 2>/dev/null | head -8
for i in 1 2 3 4; do
  if [ $((i % 2)) = 0 ]; then export LD_LIBRARY_PATH=$TL; v=torchlib; else unset LD_LIBRARY_PATH; v=rocmlib; fi
  s=$(date +%s.%N)
  NCCL_DEBUG=INFO timeout -s KILL 25 /tmp/test_shim rccl > /tmp/out_$i.txt 2> /tmp/err_$i.txt; rc=$?
  e=$(date +%s.%N)
  echo "iter $i $v rc=$rc secs=$(python3 -c "print(round($e-$s,1))") last: $(tail -1 /tmp/err_$i.txt | cut -c1-100)"
  grep -i "nccl\|rccl" /tmp/out_$i.txt | tail -6 | cut -c1-220
done
