#!/bin/bash
# TEST INFRASTRUCTURE ONLY.
# Builds oracle/_ref/libgyref.so: the reference's OWN GY_HISTOGRAM / bucket-hash / jhash / IP_PORT / wire-struct code compiled from the
# sources where they lie under /root/reference (SURVEY.md 8c recipe).  The reference's build system is not run (it needs
# folly, liburcu, boost ... none of which exist here); only header-only code that is self-contained is used.
#
# Nothing from the reference is copied into the repo: the two headers that quote-include unavailable third-party headers
# (gy_statistics.h -> folly TimeseriesSlabHistogram, gy_inet_inc.h -> liburcu) are copied into a throw-away mktemp dir next
# to small stub headers so the stubs win the quote-include lookup; only the resulting .so lands in oracle/_ref/.
set -euo pipefail
REF=${GY_REFERENCE_DIR:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/common" ]; then
	echo "build_ref.sh: $REF not present (GPU box?) - keeping prebuilt oracle/_ref if any" >&2
	exit 0
fi
mkdir -p "$OUT"
T="$(mktemp -d)"
trap 'rm -rf "$T"' EXIT
cp "$REF/common/gy_statistics.h" "$REF/common/gy_inet_inc.h" "$T/"
# liburcu is absent: the only thing gy_inet_inc.h needs from gy_rcu_inc.h is this member-injection macro
echo '#define RCU_HASH_CLASS_MEMBERS(...)' > "$T/gy_rcu_inc.h"
: > "$T/TimeseriesSlabHistogram-defs.h"
cat > "$T/gy_print_offload.h" <<'EOF'
#pragma once
#include "gy_common_inc.h"
#define ERRORPRINTCOLOR_OFFLOAD(c,fmt,...) ERRORPRINTCOLOR(c,fmt,##__VA_ARGS__)
EOF
# folly is absent: TIME_HISTOGRAM can then be declared but not instantiated (windowed parity is unpinned)
cat > "$T/TimeseriesSlabHistogram.h" <<'EOF'
#pragma once
#include <chrono>
namespace folly {
template <typename D = std::chrono::seconds> struct LegacyStatsClock { using duration = D; using time_point = std::chrono::time_point<LegacyStatsClock, D>; };
template <typename T, typename CT = LegacyStatsClock<std::chrono::seconds>> class MultiLevelTimeSeries;
template <typename T, typename H, typename CT = LegacyStatsClock<std::chrono::seconds>, typename C = MultiLevelTimeSeries<T, CT>> class TimeseriesSlabHistogram;
}
EOF
# thirdparty/SlabHistogramBucket.h (the in-tree, modified folly histogram-bucket container whose getPercentileBucketIdx is the
# percentile rule of TIME_HISTOGRAM::get_stats) is compiled where it lies; it only wants two folly headers for SCOPE_EXIT and
# glog's CHECK macros, stubbed here
mkdir -p "$T/folly"
: > "$T/folly/Conv.h"
cat > "$T/folly/ScopeGuard.h" <<'EOF'
#pragma once
#include <utility>
namespace gy_stub {
template <class F> struct ScopeExit { F f; ~ScopeExit() { f(); } };
struct ScopeExitTag {};
template <class F> ScopeExit<F> operator+(ScopeExitTag, F &&f) { return ScopeExit<F>{std::forward<F>(f)}; }
}
#define GY_STUB_CAT2(a, b) a##b
#define GY_STUB_CAT(a, b) GY_STUB_CAT2(a, b)
#define SCOPE_EXIT auto GY_STUB_CAT(gy_stub_scope_exit_, __LINE__) = gy_stub::ScopeExitTag{} + [&]()
#define CHECK_GE(a, b) ((void)0)
#define CHECK_LE(a, b) ((void)0)
EOF
# the wire structs and their validators (common/gy_comm_proto.h / .cc) compile once gy_sys_hardware.h -- which drags in libmnl,
# folly and boost for things the wire structs do not use -- is replaced by a stand-in that only declares the 16-byte GY_MACHINE_ID
# (layout of common/gy_sys_hardware.h:20-24: one std::pair<uint64_t, uint64_t>)
cp "$REF/common/gy_comm_proto.h" "$REF/common/gy_comm_proto.cc" "$T/"
cat > "$T/gy_sys_hardware.h" <<'EOF'
#pragma once
#include "gy_common_inc.h"
#include "jhash.h"
namespace gyeeta {
class GY_MACHINE_ID {
public:
	std::pair<uint64_t, uint64_t> machid_ {};
	GY_MACHINE_ID() noexcept = default;
	GY_MACHINE_ID(uint64_t hi, uint64_t lo) noexcept : machid_(hi, lo) {}
	uint64_t get_first() const noexcept { return machid_.first; }
	uint64_t get_second() const noexcept { return machid_.second; }
	uint32_t get_hash() const noexcept { return jhash2((uint32_t *)&machid_, sizeof(machid_) / sizeof(uint32_t), 0xceedfead); }
	bool operator==(const GY_MACHINE_ID &o) const noexcept { return machid_ == o.machid_; }
};
}
EOF
# LISTEN_SUMM_STATS<T> (server/gy_msocket.h:839-883) and CLUSTER_STATE_ONE (server/gy_mconnhdlr.cc:16032-16050) are small self-contained
# classes inside files that cannot be compiled here (folly, liburcu, boost, Postgres).  Their TEXT is cut out of the reference where it
# lies into the throw-away dir -- from the class head to the first "};" at the start of a line -- and compiled by ref_glue.cc
# (#if __has_include): the reference's own update() / update_from_state(), pinned without the files around them.
awk '/^template <typename T = int>/{hold=$0; next} /^class LISTEN_SUMM_STATS/{print hold; on=1} on{print} on&&/^};/{exit} {hold=""}' "$REF/server/gy_msocket.h" > "$T/ref_listen_summ_stats.h"
awk '/^struct CLUSTER_STATE_ONE : public comm::MS_CLUSTER_STATE::STATE_ONE/{on=1} on{print} on&&/^};/{exit}' "$REF/server/gy_mconnhdlr.cc" > "$T/ref_cluster_state_one.h"
grep -q "void update(const comm::LISTENER_STATE_NOTIFY" "$T/ref_listen_summ_stats.h" || rm -f "$T/ref_listen_summ_stats.h" "$T/ref_cluster_state_one.h"
grep -q "update_from_state" "$T/ref_cluster_state_one.h" 2>/dev/null || rm -f "$T/ref_listen_summ_stats.h" "$T/ref_cluster_state_one.h"
# the forced includes paper over gcc-8 era transitive-include assumptions in gy_common_inc.h; -fno-strict-aliasing -fno-strict-overflow are
# the reference's own COMMFLAGS (Makefile.common:42)
g++ -std=c++17 -O2 -D_GNU_SOURCE -DNDEBUG -DTASK_COMM_LEN=16 -pthread -fno-strict-aliasing -fno-strict-overflow -fPIC -shared -w \
	-include string -include string_view -include optional -include vector -include algorithm -include functional \
	-include chrono -include array -include tuple -include utility \
	-I"$T" -I"$REF/common" -I"$REF/thirdparty" "$HERE/ref_glue.cc" "$T/gy_comm_proto.cc" -o "$OUT/libgyref.so"
echo "built $OUT/libgyref.so"
