#!/bin/bash
# The emulated kernels (tests/cpp/kemu) under a sanitizer: tools/kemu_tsan.sh [thread|address]   (default thread)
#   thread:  LDS / global accesses of a kernel that are ordered neither by a barrier nor by an atomic show up as data races.  Expected
#            reports, all idempotent by construction: the read-before-atomicMax of the HLL registers in k_resp_host / k_conn_ingest and
#            the read-before-atomicOr of CONN_BITMAP words in k_huge_count / k_digest_huge (a register / word only grows: a stale read
#            costs at most a redundant atomic; likewise k_resp_host's re-read of the register file for the next tile's floor while other
#            workgroups raise registers: a stale floor is only lower), and same-value stores by several threads (finalize_one: the host's
#            spill stamp and the batch's `hot` flag; k_huge_merge: s_over = 1; k_resp_host's parking entry behind a tile image: every place that kept nothing stores the
#            same {dropped word, key 0} there); the read-before-atomicMax of a histogram's max_val_seen (hist_add_atomic); k_wire_round's mark[]
#            (a slot marked DURING a doubling round may already pass its mark on in that round: marks only grow and everything that gets
#            marked is a true record start of the chain, so a round can only run ahead); k_conn_ingest's relaxed read of its LDS table's
#            fill count beside the atomic adds to it (a stale count only changes how many probes a record tries).  Round 3's one REAL
#            report -- k_lstate_ingest's 96-byte store of a listener's kept state when two records of one call name the same listener --
#            is gone since round 4 (64-bit atomicMax claim + k_lstate_keep: the owner of the claim stores).
#            ~12 minutes on 8 cores.
#   address: an index past the end of a __shared__ array (a function-local static here, red zones around it), of the dynamic LDS block
#            or of a global buffer aborts the program.  Expected: no report.  ~6 minutes.
SAN=${1:-thread}
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
python -c "from oracle import oracle; oracle.lib()" || exit 1
for t in bins resp spill conn cms wire lstate topn rollup svcquery ldecide; do
	g++ -std=c++20 -O1 -g -w -fsanitize=$SAN -DKEMU_NB=4 -Itests/cpp/kemu tests/cpp/kemu/test_$t.cc -o /tmp/kemu_${t}_$SAN -Loracle -l:liboracle.so -Wl,-rpath,$R/oracle -pthread || exit 1
	TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ASAN_OPTIONS="detect_leaks=0" timeout 2400 /tmp/kemu_${t}_$SAN > /tmp/kemu_${t}_$SAN.log 2>&1
	echo "== $t"; grep -E "SUMMARY|ERROR: AddressSanitizer|kemu $t ok|FAIL" /tmp/kemu_${t}_$SAN.log | sort | uniq -c
done
# the instances the default build of a program does not reach (round 5): merges of 2048 / 4096 values, bound-address listeners, IPv6 events (also in the
# split form), a larger digest buffer, predicted runs
variant() { name=$1; src=$2; arg=$3; shift 3
	g++ -std=c++20 -O1 -g -w -fsanitize=$SAN -Itests/cpp/kemu "$@" tests/cpp/kemu/$src -o /tmp/kemu_v_${name}_$SAN -Loracle -l:liboracle.so -Wl,-rpath,$R/oracle -pthread || exit 1
	TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ASAN_OPTIONS="detect_leaks=0" timeout 2400 /tmp/kemu_v_${name}_$SAN $arg > /tmp/kemu_v_${name}_$SAN.log 2>&1
	echo "== $name"; grep -E "SUMMARY|ERROR: AddressSanitizer|kemu .* ok|FAIL" /tmp/kemu_v_${name}_$SAN.log | sort | uniq -c
}
variant bins_2048 test_bins.cc 12345 -DKEMU_BINS_VPT=8
variant bins_4096 test_bins.cc 12345 -DKEMU_BINS_VPT=16
variant bins_16384_streamed test_bins.cc 12345 -DKEMU_BINS_VPT=64
variant resp_bound_address test_resp.cc 4243 -DKEMU_TPT=16 -DKEMU_MODE=1
variant resp_ipv6 test_resp.cc 4244 -DKEMU_TPT=16 -DKEMU_MODE=2
variant resp_ipv6_split test_resp.cc 4245 -DKEMU_TPT=16 -DKEMU_MODE=2 -DKEMU_SPLIT -DKEMU_NB=4
variant resp_pend_cap_1536 test_resp.cc 4246 -DKEMU_TPT=16 -DKEMU_PEND_CAP=1536 -DKEMU_NB=12
variant spill_predicted test_spill.cc 778 -DKEMU_PRESPILL
