// TEST INFRASTRUCTURE (CPU): the LOGIC of the wire front-end's record walk -- k_wire_next / seed / round / count, the three scan kernels,
// k_wire_emit / check, launched in the order of wire_decode (gyeeta_amd/csrc/gys_engine.hip) -- under the CPU stand-in of the device
// model, on streams of TCP_CONN_NOTIFY and LISTENER_STATE_NOTIFY messages whose length fields, record counts and message lengths are
// partly CORRUPTED: the kernels say "malformed" exactly when the oracle's restatement of the reference's validators
// (gyo_tcp_conn_validate / gyo_listener_state_validate: common/gy_comm_proto.cc:840-881, :955-996; pinned against the reference's own
// code in tests/test_wire.py) rejects a message, and for streams it accepts the offset list equals the serial p += get_elem_size()
// walk.  These bytes come from the network: built with -fsanitize=address (tools/kemu_tsan.sh address) the run also shows that no
// corrupted length makes a kernel index outside the stream or its work arrays (every buffer here is sized exactly as the library sizes it).
// Build + run: tests/test_kernel_logic_cpu.py.
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+r"(a[0]), "+r"(a[1]), "+r"(a[2]), "+r"(a[3]))
#define GYS_DYN_LDS(type, name) type *name = (type *)kemu::dyn_lds()
#include "../../../gyeeta_amd/csrc/gys_kernels.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <random>

#include "../../../oracle/gy_oracle.h"

using namespace gys;

namespace {
int fails = 0;
#define CHECK(c, ...)                                               \
	do {                                                        \
		if (!(c)) {                                         \
			if (fails++ < 20) {                         \
				printf("FAIL %s:%d: ", __FILE__, __LINE__); \
				printf(__VA_ARGS__);                \
				printf("\n");                       \
			}                                           \
		}                                                   \
	} while (0)
void put32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }
void put16(uint8_t *p, uint16_t v) { memcpy(p, &v, 2); }
uint32_t get32(const uint8_t *p)
{
	uint32_t v;
	memcpy(&v, p, 4);
	return v;
}

struct Msg {
	size_t pos;      // of the COMM_HEADER in the stream
	uint32_t nrec;   // records actually laid down
};
} // namespace

int main(int argc, char **argv)
{
	if (!kemu::can_run(256u)) {
		printf("kemu: this process cannot have 256 threads\n");
		return 77;
	}
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 11u);
	const int NSTREAMS = argc > 2 ? atoi(argv[2]) : 60;
	int n_ok = 0, n_bad = 0;
	for (int it = 0; it < NSTREAMS; ++it) {
		const uint32_t kind = it & 1u; // 0 = TCP_CONN_NOTIFY, 1 = LISTENER_STATE_NOTIFY (wire_decode takes the messages of one kind at a time)
		const uint32_t fixed = kind == 0 ? 280u : 88u;
		const uint32_t nmsg = 1u + rng() % 4u;
		std::vector<uint8_t> raw;
		std::vector<Msg> layout;
		for (uint32_t m = 0; m < nmsg; ++m) {
			const uint32_t nrec = rng() % 13u; // (a message may announce no records)
			Msg mm{raw.size(), nrec};
			raw.resize(raw.size() + 24, 0);
			for (uint32_t r = 0; r < nrec; ++r) {
				const uint32_t var = kind == 0 ? rng() % 41u : rng() % 30u;
				const uint32_t pad = (8u - var % 8u) % 8u;
				const size_t at = raw.size();
				raw.resize(at + fixed + var + pad);
				for (size_t k = at; k < raw.size(); ++k) raw[k] = (uint8_t)rng();
				if (kind == 0) {
					put16(&raw[at + 272], (uint16_t)var);
					raw[at + 279] = (uint8_t)pad;
				} else {
					raw[at + 85] = (uint8_t)var;
					raw[at + 86] = (uint8_t)pad;
				}
			}
			const uint32_t hpad = rng() % 3u ? 0u : rng() % 8u; // COMM_HEADER::padding_sz_ < 8: bytes of the (8-byte multiple) total_sz_ that are not data
			const uint32_t total = (uint32_t)(raw.size() - mm.pos) + (hpad ? 8u : 0u);
			raw.resize(mm.pos + total, 0);
			put32(&raw[mm.pos], 0x05666605u);
			put32(&raw[mm.pos + 4], total);
			put32(&raw[mm.pos + 8], 14u);
			put32(&raw[mm.pos + 12], hpad);
			put32(&raw[mm.pos + 16], kind == 0 ? 0x30Cu : 0x309u);
			put32(&raw[mm.pos + 20], nrec);
			layout.push_back(mm);
		}
		// ---- corruption (two streams in three): a length field, a padding field, the record count, or the message's padding_sz_
		const uint32_t ncorrupt = (it % 3 == 0) ? 0u : 1u + rng() % 3u;
		for (uint32_t c = 0; c < ncorrupt; ++c) {
			const Msg &mm = layout[rng() % nmsg];
			const uint32_t what = rng() % 5u;
			if (what == 0) { // more (or fewer) records announced than present
				put32(&raw[mm.pos + 20], rng() % 2u ? mm.nrec + 1u + rng() % 3u : (mm.nrec ? rng() % mm.nrec : 0u));
			} else if (what == 1) { // padding_sz_ of the header: the last record may no longer fit
				put32(&raw[mm.pos + 12], rng() % 8u);
			} else if (mm.nrec) { // a length or padding byte of some record: every later record start of the message moves
				size_t at = mm.pos + 24;
				const uint32_t target = rng() % mm.nrec;
				for (uint32_t r = 0; r < target; ++r) at += gyo_tcp_conn_elem_size(&raw[at]) * (kind == 0) + gyo_listener_state_elem_size(&raw[at]) * (kind == 1);
				if (at + fixed > raw.size()) continue; // (an earlier corruption already moved the chain out of the stream)
				if (kind == 0) {
					if (what == 2) put16(&raw[at + 272], (uint16_t)(rng() % 3u ? rng() % 600u : 0xFFFFu));
					else raw[at + 279] = (uint8_t)rng();
				} else {
					if (what == 2) raw[at + 85] = (uint8_t)rng();
					else raw[at + 86] = (uint8_t)rng();
				}
			}
		}
		// ---- the stream as the library holds it: 8-byte aligned copy, nslots = nbytes / 8 + 1 (one terminal slot)
		const uint64_t nbytes = raw.size();
		std::vector<uint64_t> buf(nbytes / 8); // EXACTLY the stream: a read past it is an out-of-bounds access
		memcpy(buf.data(), raw.data(), nbytes);
		const uint8_t *sb = (const uint8_t *)buf.data();
		std::vector<WireMsg> msgs;
		uint32_t nrec_total = 0, maxev = 1;
		bool oracle_ok = true, host_rejects = false;
		for (const Msg &mm : layout) {
			const uint32_t total = get32(sb + mm.pos + 4), padding = get32(sb + mm.pos + 12), nevents = get32(sb + mm.pos + 20);
			const uint32_t act = total - padding;
			if (act < 16u + 8u) { // COMM_HEADER::validate (common/gy_comm_proto.cc:10-57) on the host, before any kernel: gys_ingest_comm_stream returns here
				host_rejects = true;
				CHECK(!(kind == 0 ? gyo_tcp_conn_validate(sb + mm.pos) : gyo_listener_state_validate(sb + mm.pos)), "stream %d: a message shorter than its headers passes the validator", it);
				break;
			}
			WireMsg w{};
			w.pay_slot = (uint32_t)(mm.pos + 24) / 8u;
			w.end_slot = (uint32_t)(mm.pos + act) / 8u;
			w.nevents = nevents;
			w.out_base = nrec_total;
			w.kind = kind;
			nrec_total += nevents;
			maxev = std::max(maxev, nevents);
			msgs.push_back(w);
			oracle_ok = oracle_ok && (kind == 0 ? gyo_tcp_conn_validate(sb + mm.pos) : gyo_listener_state_validate(sb + mm.pos));
		}
		if (host_rejects) {
			++n_bad;
			continue;
		}
		const uint32_t nslots = (uint32_t)(nbytes / 8) + 1u, nmsgs = (uint32_t)msgs.size();
		std::vector<uint32_t> jump[2] = {std::vector<uint32_t>(nslots), std::vector<uint32_t>(nslots)}, cnt(nslots), rank(nslots), offsets(std::max(nrec_total, 1u), 0xFFFFFFFFu);
		const uint32_t nblk = (nslots + GYS_SCAN_TILE - 1) / GYS_SCAN_TILE;
		std::vector<uint32_t> bsums(nblk + 2), status(4, 0);
		std::vector<uint8_t> mark(nslots, 0), flags(nslots);
		const uint32_t gs = (nslots + 255u) / 256u, gm = (nmsgs + 255u) / 256u;
		kemu::launch(gs, 256, 0, [&] { k_wire_next(buf.data(), msgs.data(), nmsgs, nslots, jump[0].data(), flags.data()); });
		kemu::launch(gm, 256, 0, [&] { k_wire_seed(msgs.data(), nmsgs, mark.data()); });
		int cur = 0;
		for (uint32_t span = 1; span < maxev; span <<= 1) {
			kemu::launch(gs, 256, 0, [&] { k_wire_round(nslots, jump[cur].data(), jump[cur ^ 1].data(), mark.data()); });
			cur ^= 1;
		}
		kemu::launch(gs, 256, 0, [&] { k_wire_count(nslots, mark.data(), flags.data(), cnt.data()); });
		kemu::launch(nblk, 256, 0, [&] { k_scan_block_sums(cnt.data(), nslots, bsums.data()); });
		kemu::launch(1, 256, 0, [&] { k_scan_top(bsums.data(), nblk); });
		kemu::launch(nblk, 256, 0, [&] { k_scan_final(cnt.data(), nslots, bsums.data(), rank.data()); });
		kemu::launch(gs, 256, 0, [&] { k_wire_emit(msgs.data(), nmsgs, nslots, cnt.data(), rank.data(), flags.data(), offsets.data(), status.data()); });
		kemu::launch(gm, 256, 0, [&] { k_wire_check(msgs.data(), nmsgs, nslots, cnt.data(), rank.data(), status.data()); });

		CHECK((status[0] == 0) == oracle_ok, "stream %d (kind %u, %u corruptions): kernels say status %u, the validators say %s", it, kind, ncorrupt, status[0],
		      oracle_ok ? "well-formed" : "malformed");
		if (oracle_ok) {
			++n_ok;
			for (uint32_t m = 0; m < nmsgs; ++m) { // the serial walk of the reference's loops (server/gy_mconnhdlr.cc:9130, :11175)
				size_t at = layout[m].pos + 24;
				for (uint32_t r = 0; r < msgs[m].nevents; ++r) {
					CHECK(offsets[msgs[m].out_base + r] == at, "stream %d message %u record %u: offset %u, serial walk %zu", it, m, r, offsets[msgs[m].out_base + r], at);
					at += kind == 0 ? gyo_tcp_conn_elem_size(sb + at) : gyo_listener_state_elem_size(sb + at);
				}
			}
		} else
			++n_bad;
	}
	if (fails) {
		printf("kemu wire: %d FAILURES\n", fails);
		return 1;
	}
	printf("kemu wire ok: %d streams accepted with the serial walk's offsets, %d rejected as the validators reject them\n", n_ok, n_bad);
	return n_ok >= NSTREAMS / 4 && n_bad >= NSTREAMS / 6 ? 0 : 2; // (both outcomes must have been exercised)
}
