// TEST INFRASTRUCTURE (CPU): the LOGIC of the window's Count-Min rows -- k_cms_partial (eight services per thread and round, column
// filter by row half, LDS image) and k_cms_reduce -- under the CPU stand-in of the device model: the arena after two windows equals the
// oracle's gyo_cms_add of every service's event count.  Service counts chosen so that a chunk ends inside a round of eight (ragged
// tail), is shorter than one round, and is empty.  Build + run: tests/test_kernel_logic_cpu.py.
#define GYS_OPAQUE_VGPR(x) asm volatile("" : "+r"(x))
#define GYS_OPAQUE_LOADED4(a) asm volatile("" : "+r"(a[0]), "+r"(a[1]), "+r"(a[2]), "+r"(a[3]))
#define GYS_DYN_LDS(type, name) type *name = (type *)kemu::dyn_lds()
#include "../../../gyeeta_amd/csrc/gys_kernels.hpp"

#include <stdio.h>
#include <stdlib.h>

#include <random>

#include "../../../oracle/gy_oracle.h"

using namespace gys;

int main(int argc, char **argv)
{
	if (!kemu::can_run(1024u)) {
		printf("kemu: this process cannot have 1024 threads\n");
		return 77;
	}
	std::mt19937 rng(argc > 1 ? (unsigned)atoi(argv[1]) : 5u);
	const uint32_t NCMS = GYS_CMS_D * GYS_CMS_W;
	int fails = 0;
	std::vector<uint32_t> arena(NCMS, 0), want(NCMS, 0);
	struct Case { uint32_t nsvc, nch; };
	const Case cases[] = {{19001, 2}, {1, 2}, {8192 + 1024 + 17, 1}};
	uint64_t total = 0;
	for (const Case &c : cases) {
		std::vector<uint32_t> resp_win(c.nsvc);
		std::vector<uint64_t> gid(c.nsvc);
		for (uint32_t s = 0; s < c.nsvc; ++s) {
			gid[s] = ((uint64_t)rng() << 32) | rng();
			resp_win[s] = (rng() % 3u) ? 1u + rng() % 100000u : 0u; // a third of the services idle in the window
			if (resp_win[s]) {
				const uint32_t gw[2] = {(uint32_t)gid[s], (uint32_t)(gid[s] >> 32)};
				gyo_cms_add(want.data(), gw, 2, resp_win[s]);
				total += resp_win[s];
			}
		}
		std::vector<uint32_t> partial((size_t)c.nch * NCMS, 0xDEADBEEFu); // (every cell of every partial row is written by its workgroup)
		for (uint32_t y = 0; y < GYS_CMS_D * 2u; ++y)
			kemu::launch(c.nch, 1024, GYS_CMSF_CELLS * 4u, [&] {
				kemu::t_blockIdx.y = y;
				k_cms_partial(resp_win.data(), gid.data(), c.nsvc, c.nch, partial.data());
			});
		kemu::launch((NCMS + 255u) / 256u, 256, 0, [&] { k_cms_reduce(partial.data(), c.nch, arena.data()); });
		for (uint32_t k = 0; k < NCMS; ++k)
			if (arena[k] != want[k] && fails++ < 10) printf("FAIL nsvc %u: cell %u is %u, oracle %u\n", c.nsvc, k, arena[k], want[k]);
	}
	if (fails) {
		printf("kemu cms: %d FAILURES\n", fails);
		return 1;
	}
	printf("kemu cms ok: %llu events of %zu windows in the rows\n", (unsigned long long)total, sizeof(cases) / sizeof(cases[0]));
	return 0;
}
