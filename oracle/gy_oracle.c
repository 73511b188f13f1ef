/*
 * TEST INFRASTRUCTURE ONLY -- see gy_oracle.h.  Plain C (gcc), no dependencies.
 * Every function cites the reference file:line (under /root/reference) it follows.
 */
#include "gy_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

/* ================================================================ hashing */

/* common/jhash.h:23-34 __jhash_mix */
#define GYO_MIX(a, b, c)            \
	do {                        \
		a -= b; a -= c; a ^= (c >> 13); \
		b -= c; b -= a; b ^= (a << 8);  \
		c -= a; c -= b; c ^= (b >> 13); \
		a -= b; a -= c; a ^= (c >> 12); \
		b -= c; b -= a; b ^= (a << 16); \
		c -= a; c -= b; c ^= (b >> 5);  \
		a -= b; a -= c; a ^= (c >> 3);  \
		b -= c; b -= a; b ^= (a << 10); \
		c -= a; c -= b; c ^= (b >> 15); \
	} while (0)

#define GYO_GOLDEN 0x9e3779b9u /* common/jhash.h:37 */
#define GYO_SEED 0xceedfeadu   /* common/gy_common_inc.h:1112 */

/* common/jhash.h:43-83 */
uint32_t gyo_jhash(const void *key, uint32_t length, uint32_t initval)
{
	uint32_t a, b, c, len = length;
	const uint8_t *k = (const uint8_t *)key;

	a = b = GYO_GOLDEN;
	c = initval;
	while (len >= 12) {
		a += (k[0] + ((uint32_t)k[1] << 8) + ((uint32_t)k[2] << 16) + ((uint32_t)k[3] << 24));
		b += (k[4] + ((uint32_t)k[5] << 8) + ((uint32_t)k[6] << 16) + ((uint32_t)k[7] << 24));
		c += (k[8] + ((uint32_t)k[9] << 8) + ((uint32_t)k[10] << 16) + ((uint32_t)k[11] << 24));
		GYO_MIX(a, b, c);
		k += 12;
		len -= 12;
	}
	c += length;
	switch (len) { /* all cases fall through */
	case 11: c += ((uint32_t)k[10] << 24); /* FALLTHRU */
	case 10: c += ((uint32_t)k[9] << 16);  /* FALLTHRU */
	case 9: c += ((uint32_t)k[8] << 8);    /* FALLTHRU */
	case 8: b += ((uint32_t)k[7] << 24);   /* FALLTHRU */
	case 7: b += ((uint32_t)k[6] << 16);   /* FALLTHRU */
	case 6: b += ((uint32_t)k[5] << 8);    /* FALLTHRU */
	case 5: b += k[4];                     /* FALLTHRU */
	case 4: a += ((uint32_t)k[3] << 24);   /* FALLTHRU */
	case 3: a += ((uint32_t)k[2] << 16);   /* FALLTHRU */
	case 2: a += ((uint32_t)k[1] << 8);    /* FALLTHRU */
	case 1: a += k[0];
	}
	GYO_MIX(a, b, c);
	return c;
}

/* common/jhash.h:88-113 */
uint32_t gyo_jhash2(const uint32_t *k, uint32_t length, uint32_t initval)
{
	uint32_t a, b, c, len = length;

	a = b = GYO_GOLDEN;
	c = initval;
	while (len >= 3) {
		a += k[0];
		b += k[1];
		c += k[2];
		GYO_MIX(a, b, c);
		k += 3;
		len -= 3;
	}
	c += length * 4;
	switch (len) {
	case 2: b += k[1]; /* FALLTHRU */
	case 1: a += k[0];
	}
	GYO_MIX(a, b, c);
	return c;
}

/* common/jhash.h:122-140 */
uint32_t gyo_jhash_3words(uint32_t a, uint32_t b, uint32_t c, uint32_t initval)
{
	a += GYO_GOLDEN;
	b += GYO_GOLDEN;
	c += initval;
	GYO_MIX(a, b, c);
	return c;
}
uint32_t gyo_jhash_2words(uint32_t a, uint32_t b, uint32_t initval) { return gyo_jhash_3words(a, b, 0, initval); }
uint32_t gyo_jhash_1word(uint32_t a, uint32_t initval) { return gyo_jhash_3words(a, 0, 0, initval); }

/* common/gy_common_inc.h:1110-1123 */
uint32_t gyo_get_uint32_hash(uint32_t k) { return gyo_jhash_1word(k, GYO_SEED); }
uint32_t gyo_get_uint64_hash(uint64_t k) { return gyo_jhash_2words((uint32_t)(k & 0xFFFFFFFFu), (uint32_t)(k >> 32), GYO_SEED); }

/* GY_IP_ADDR as the reference builds it from raw address bytes: set_ip(uint32_t) common/gy_common_inc.h:10673-10679 (ip128_be_ = 0,
 * ip32_be_ = the address) / set_ip(unsigned __int128) :10686-10692 (ip32_be_ = 0, then get_ipv6_type_flags() :11040-11129).  ip32_be_ and
 * embedded_ipv4_ are ONE storage (the union at :10497-10500), and get_ipv6_type_flags stores the IPv4 address an IPv6 address embeds into
 * embedded_ipv4_: 2002::/16 (6to4, bytes 2..5 :11064-11069), ::ffff:a.b.c.d (bytes 12..15 :11079-11094), 64:ff9b::/32 (NAT64, bytes 12..15
 * :11096-11104) -- checked in that order, after :: (:11047-11050) and ::1 (:11052-11060), and an address of 2000::/4 that is not 2002::/16
 * returns before the later checks (:11062-11077).  Such an address therefore carries ip32_be_ = the embedded IPv4 address.
 * Out: *ip32 = ip32_be_ (as the 4 address bytes read as a native u32), ip128 = the 16 bytes of ip128_be_.  Returns is_any_address()
 * (:10915-10918: ipflags_ & (IPv4_ANY | IPv6_ANY), i.e. 0.0.0.0 :11135-11138 or ::). */
int gyo_ip_norm(const uint8_t *ip, int is_v6, uint32_t *ip32, uint8_t ip128[16])
{
	static const uint8_t zero[16];

	if (!is_v6) {
		memcpy(ip32, ip, 4);
		memset(ip128, 0, 16);
		return *ip32 == 0;
	}
	memcpy(ip128, ip, 16);
	*ip32 = 0;
	if (!memcmp(ip, zero, 16)) return 1;                                     /* IPv6_ANY */
	if (!memcmp(ip, zero, 15) && ip[15] == 1) return 0;                      /* ::1 */
	if ((ip[0] & 0xF0) == 0x20) {
		if (ip[0] == 0x20 && ip[1] == 0x02) memcpy(ip32, ip + 2, 4);     /* 2002:: */
		return 0;
	}
	if (!memcmp(ip, zero, 8) && ip[8] == 0 && ip[9] == 0 && ip[10] == 0xFF && ip[11] == 0xFF) { /* ::ffff:1.2.3.4 */
		memcpy(ip32, ip + 12, 4);
		return 0;
	}
	if (ip[0] == 0 && ip[1] == 0x64 && ip[2] == 0xFF && ip[3] == 0x9B) memcpy(ip32, ip + 12, 4); /* 64:ff9b:: */
	return 0;
}

/* GY_IP_ADDR::operator== common/gy_common_inc.h:10629-10636: by ip32_be_ when either side has one, else by the 16 bytes */
int gyo_ip_equal(uint32_t a32, const uint8_t a128[16], uint32_t b32, const uint8_t b128[16])
{
	if (a32 || b32) return a32 == b32;
	return memcmp(a128, b128, 16) == 0;
}

/* GY_IP_ADDR::get_as_inaddr common/gy_common_inc.h:10950-10959: ip32_be_ != 0 -> those 4 bytes, else the 16 ip128 bytes.
 * An IPv4 address of 0.0.0.0 therefore hashes as 16 zero bytes ("IP Any Address will be considered as IPv6"), and an IPv6 address that
 * embeds an IPv4 one as that IPv4 address ("IPv4 Mapped IPv6 addresses will be returned as IPv4"). */
static int gyo_inaddr(const uint8_t *ip, int is_v6, uint8_t *buf)
{
	uint32_t ip32;
	uint8_t ip128[16];

	gyo_ip_norm(ip, is_v6, &ip32, ip128);
	if (ip32) {
		memcpy(buf, &ip32, 4);
		return 4;
	}
	memcpy(buf, ip128, 16);
	return 16;
}

/* IP_PORT::get_hash common/gy_common_inc.h:11226-11244 : [inaddr] [port>>8] [port&0xff] 00 00 */
uint32_t gyo_ip_port_words(const uint8_t *ip, int is_v6, uint16_t port, int ignore_ip, uint32_t out[5])
{
	uint8_t buf[24];
	int len = ignore_ip ? 0 : gyo_inaddr(ip, is_v6, buf);

	buf[len++] = (uint8_t)(port >> 8);
	buf[len++] = (uint8_t)(port & 0xFF);
	buf[len++] = 0;
	buf[len++] = 0;
	memcpy(out, buf, (size_t)len);
	return (uint32_t)len / 4;
}

uint32_t gyo_ip_port_hash(const uint8_t *ip, int is_v6, uint16_t port, int ignore_ip)
{
	uint32_t w[5];
	uint32_t n = gyo_ip_port_words(ip, is_v6, port, ignore_ip, w);
	return gyo_jhash2(w, n, GYO_SEED);
}

/* NS_IP_PORT::get_hash common/gy_inet_inc.h:136-160 : [inaddr] [port native LE] 00 00 [inode 8 bytes] */
uint32_t gyo_ns_ip_port_hash(const uint8_t *ip, int is_v6, uint16_t port, uint64_t inode, int ignore_ip)
{
	uint8_t buf[32];
	uint32_t w[8];
	int len = ignore_ip ? 0 : gyo_inaddr(ip, is_v6, buf);

	memcpy(buf + len, &port, 2);
	len += 2;
	buf[len++] = 0;
	buf[len++] = 0;
	memcpy(buf + len, &inode, 8);
	len += 8;
	memcpy(w, buf, (size_t)len);
	return gyo_jhash2(w, (uint32_t)len / 4, GYO_SEED);
}

/* PAIR_IP_PORT::get_hash common/gy_inet_inc.h:225-247 : [cli inaddr][cli port LE]00 00[ser inaddr][ser port LE]00 00 */
/* the same from two GY_IP_ADDR OBJECTS (their ip32_be_ and ip128_be_ members as they stand): what get_as_inaddr reads */
uint32_t gyo_pair_words_obj(uint32_t c32, const uint8_t c128[16], uint16_t cport, uint32_t s32, const uint8_t s128[16], uint16_t sport, uint32_t out[10])
{
	uint8_t buf[48];
	int len = 0;

	if (c32) { memcpy(buf, &c32, 4); len = 4; } else { memcpy(buf, c128, 16); len = 16; }
	memcpy(buf + len, &cport, 2);
	len += 2;
	buf[len++] = 0;
	buf[len++] = 0;
	if (s32) { memcpy(buf + len, &s32, 4); len += 4; } else { memcpy(buf + len, s128, 16); len += 16; }
	memcpy(buf + len, &sport, 2);
	len += 2;
	buf[len++] = 0;
	buf[len++] = 0;
	memcpy(out, buf, (size_t)len);
	return (uint32_t)len / 4;
}

uint32_t gyo_pair_ip_port_words(const uint8_t *cip, int c6, uint16_t cport, const uint8_t *sip, int s6, uint16_t sport,
				uint32_t out[10])
{
	uint32_t c32, s32;
	uint8_t c128[16], s128[16];

	gyo_ip_norm(cip, c6, &c32, c128); /* the objects as the reference builds them from raw address bytes */
	gyo_ip_norm(sip, s6, &s32, s128);
	return gyo_pair_words_obj(c32, c128, cport, s32, s128, sport, out);
}

uint32_t gyo_pair_ip_port_hash(const uint8_t *cip, int c6, uint16_t cport, const uint8_t *sip, int s6, uint16_t sport)
{
	uint32_t w[10];
	uint32_t n = gyo_pair_ip_port_words(cip, c6, cport, sip, s6, sport, w);
	return gyo_jhash2(w, n, GYO_SEED);
}

/* GY_MACHINE_ID::get_hash common/gy_sys_hardware.h:82-85 : jhash2 over std::pair<uint64,uint64> (first, second) */
uint32_t gyo_machine_id_hash(uint64_t first, uint64_t second)
{
	uint32_t w[4];
	memcpy(w, &first, 8);
	memcpy(w + 2, &second, 8);
	return gyo_jhash2(w, 4, GYO_SEED);
}

/* the engine's 64-bit sketch hash built from two reference jhash2 passes */
uint64_t gyo_hash64(const uint32_t *words, uint32_t nwords)
{
	return ((uint64_t)gyo_jhash2(words, nwords, GYO_SEED) << 32) | (uint64_t)gyo_jhash2(words, nwords, GYO_GOLDEN);
}

/* ================================================================ bucket hashes (common/gy_statistics.h:1565-2063) */

typedef struct {
	int nthr;            /* GY_ARRAY_SIZE(nthresholds) */
	int64_t thr[14];
	int is_fixed_diff;   /* FIXED_DIFF_HASH :1584-1622 */
	int64_t fd_min, fd_maxp1, fd_diff;
	int arg_bits;        /* width of the get_bucket_from_data() parameter: 64, 32 or 8 (narrowing happens at the call) */
	int t_bits;          /* width of the GY_HISTOGRAM<T,...> T the reference instantiates with this hash */
} gyo_hash_def;

static const gyo_hash_def g_defs[GYO_NKINDS] = {
	/* RESP_TIME_HASH :1674-1726 ; GY_HISTOGRAM<int64_t,...> test/test_histogram.cc:11 */
	{13, {1, 10, 30, 60, 100, 150, 200, 300, 450, 700, 1000, 3000, 15000}, 0, 0, 0, 0, 64, 64},
	/* SEMI_LOG_HASH :1729-1780 (takes int) */
	{12, {1, 10, 100, 500, 1000, 5000, 25000, 50000, 100000, 300000, 1000000, 5000000}, 0, 0, 0, 0, 32, 32},
	/* SEMI_LOG_HASH_LO :1782-1833 ; qps_hist_ common/gy_socket_stat.h:633 */
	{13, {1, 10, 50, 200, 500, 1000, 3000, 6000, 10000, 15000, 25000, 60000, 150000}, 0, 0, 0, 0, 32, 32},
	/* DURATION_HASH :1835-1885 */
	{13, {1, 10, 25, 50, 125, 400, 1000, 3000, 6000, 10000, 25000, 40000, 65000}, 0, 0, 0, 0, 32, 32},
	/* HASH_10_5000 :1908-1958 */
	{12, {10, 25, 50, 75, 100, 150, 300, 500, 800, 1000, 2000, 5000}, 0, 0, 0, 0, 32, 32},
	/* HASH_5_250 :1960-2011 */
	{10, {5, 10, 20, 40, 60, 80, 100, 140, 200, 250}, 0, 0, 0, 0, 32, 32},
	/* HASH_1_3000 :2013-2063 */
	{12, {1, 5, 10, 25, 50, 75, 100, 150, 300, 500, 1000, 3000}, 0, 0, 0, 0, 32, 32},
	/* PERCENT_HASH = FIXED_DIFF_HASH<int64_t,0,100,10> :1624 ; thresholds via gy_create_threshold_array :1567-1581 */
	{11, {9, 19, 29, 39, 49, 59, 69, 79, 89, 99, 100}, 1, 0, 101, 10, 64, 32},
	/* FIXED_DIFF_HASH<int8_t,9,26,5> test/test_histogram.cc:16 */
	{4, {13, 18, 23, 26}, 1, 9, 27, 5, 8, 8},
	/* FIXED_DIFF_HASH<int,-15,-3,4> test/test_histogram.cc:88 */
	{4, {-12, -8, -4, -3}, 1, -15, -2, 4, 32, 32},
};

int gyo_hist_nbuckets(int kind) { return g_defs[kind].nthr + 2; }

static int64_t narrow(int64_t v, int bits)
{
	if (bits == 32) return (int64_t)(int32_t)v;
	if (bits == 8) return (int64_t)(int8_t)v;
	return v;
}

/* get_bucket_from_data: RESP_TIME_HASH :1698-1725 and the identical table walkers :1748-1778 etc; FIXED_DIFF :1608-1618 */
static uint32_t bucket_core(const gyo_hash_def *d, int64_t data)
{
	const int nb = d->nthr + 2;
	int64_t min_value, max_value;

	if (d->is_fixed_diff) {
		if (data < d->fd_min) return 0;
		if (data >= d->fd_maxp1) return (uint32_t)(nb - 1);
		return (uint32_t)(1 + (data - d->fd_min) / d->fd_diff);
	}
	min_value = 0;
	max_value = d->thr[d->nthr - 1] + 1;
	if (data < min_value) return 0;
	if (data >= max_value) return (uint32_t)(nb - 1);
	{
		/* mid-slot shortcut then linear "data <= nthresholds[nb] -> nb + 1" */
		const int mid = d->nthr / 2;
		int i = (data >= d->thr[mid]) ? mid : 0;
		for (; i < d->nthr; i++) {
			if (data <= d->thr[i]) return (uint32_t)(i + 1);
		}
		return (uint32_t)(i + 1);
	}
}

uint32_t gyo_bucket(int kind, int64_t data)
{
	const gyo_hash_def *d = &g_defs[kind];
	/* GY_HISTOGRAM::add_data(T data) then hash_(data): value is first narrowed to T, then to the hash's parameter type */
	return bucket_core(d, narrow(narrow(data, d->t_bits), d->arg_bits));
}

void gyo_bucket_many(int kind, const int64_t *v, size_t n, uint32_t *out)
{
	for (size_t i = 0; i < n; i++) out[i] = gyo_bucket(kind, v[i]);
}

/* get_bucket_max_threshold<HashClass,T> common/gy_statistics.h:500-515 */
int64_t gyo_bucket_max_threshold(int kind, size_t id)
{
	const gyo_hash_def *d = &g_defs[kind];
	const size_t nb = (size_t)d->nthr + 2;
	const int64_t min_value = d->is_fixed_diff ? d->fd_min : 0;
	const int64_t max_value = d->is_fixed_diff ? d->fd_maxp1 : d->thr[d->nthr - 1] + 1;

	if (id == 0) return narrow(min_value - 1, d->t_bits);
	if (id >= nb - 1) {
		int64_t maxt = d->t_bits == 64 ? LONG_MAX : (d->t_bits == 32 ? INT_MAX : SCHAR_MAX);
		int64_t lesst = max_value >= INT_MAX ? LONG_MAX : (max_value > (SHRT_MAX >> 1) ? INT_MAX : SHRT_MAX);
		return lesst < maxt ? lesst : maxt;
	}
	return narrow(d->thr[id - 1], d->t_bits);
}

/* ================================================================ GY_HISTOGRAM (common/gy_statistics.h:555-791) */

static int64_t t_min(int bits) { return bits == 64 ? LONG_MIN : (bits == 32 ? INT_MIN : SCHAR_MIN); }

void gyo_hist_init(gyo_hist *h, int kind)
{
	memset(h, 0, sizeof(*h));
	h->kind = kind;
	h->nbuckets = gyo_hist_nbuckets(kind);
	h->max_val_seen = t_min(g_defs[kind].t_bits); /* :563 std::numeric_limits<T>::min() */
}

/* add_data :596-623 (clock bookkeeping omitted: not part of the arithmetic) */
uint32_t gyo_hist_add(gyo_hist *h, int64_t data)
{
	const gyo_hash_def *d = &g_defs[h->kind];
	const int64_t v = narrow(data, d->t_bits);
	const uint32_t b = bucket_core(d, narrow(v, d->arg_bits));

	h->stats[b].sum += v; /* HIST_SERIAL::add :463-467 */
	h->stats[b].count++;
	h->total_count++;
	if (h->max_val_seen < v) h->max_val_seen = v;
	return b;
}

void gyo_hist_add_many(gyo_hist *h, const int64_t *v, size_t n)
{
	for (size_t i = 0; i < n; i++) gyo_hist_add(h, v[i]);
}

/* add_histogram :625-628 -> update_from_serialized :641-660 */
void gyo_hist_merge(gyo_hist *dst, const gyo_hist *src)
{
	for (int i = 0; i < dst->nbuckets; i++) {
		dst->stats[i].count += src->stats[i].count;
		dst->stats[i].sum += src->stats[i].sum;
	}
	dst->total_count += src->total_count;
	if (dst->max_val_seen < src->max_val_seen) dst->max_val_seen = src->max_val_seen;
}

/* get_percentiles :707-791 */
void gyo_percentiles_raw(int kind, const gyo_hist_serial *stats, uint64_t total_count, gyo_hist_data *pdata, size_t npct, float *pavg)
{
	const int nb = gyo_hist_nbuckets(kind);

	if (pavg) { /* :734-750 */
		int64_t total_sum = 0, cnt = total_count ? (int64_t)total_count : 1;
		for (int i = 0; i < nb; i++) total_sum += stats[i].sum;
		*pavg = (total_sum * 1.0f) / cnt;
	}
	for (size_t n = 0; n < npct; n++) {
		float multiplier = (float)(pdata[n].percentile / 100.0); /* float/double -> double, stored to float :757 */
		const uint64_t ncutoff = (uint64_t)((float)total_count * multiplier); /* size_t * float -> float -> size_t :758 */
		uint64_t total = 0;
		int64_t sum = 0;
		int i;

		for (i = 0; i < nb; i++) {
			total += stats[i].count;
			sum += stats[i].sum;
			if (total >= ncutoff) {
				pdata[n].count = total;
				pdata[n].data_value = gyo_bucket_max_threshold(kind, (size_t)i);
				pdata[n].sum = sum;
				break;
			}
		}
		if (i < nb) continue;
		/* :779-789 only reachable when (float)total_count*mult rounds above the bucket sum */
		pdata[n].count = total;
		pdata[n].data_value = gyo_bucket_max_threshold(kind, total_count > 0 ? (size_t)nb : 0);
		pdata[n].sum = sum;
	}
}

void gyo_hist_percentiles(const gyo_hist *h, gyo_hist_data *pdata, size_t npct, uint64_t *total, int64_t *maxv, float *pavg)
{
	if (total) *total = h->total_count;
	if (maxv) *maxv = h->max_val_seen;
	gyo_percentiles_raw(h->kind, h->stats, h->total_count, pdata, npct, pavg);
}

void gyo_keyed_hist_ingest(int kind, const uint32_t *keyidx, const int32_t *vals, size_t n, gyo_hist_serial *stats, uint64_t *total,
			   int64_t *maxv)
{
	const gyo_hash_def *d = &g_defs[kind];

	for (size_t i = 0; i < n; i++) {
		const int64_t v = narrow((int64_t)vals[i], d->t_bits);
		const uint32_t b = bucket_core(d, narrow(v, d->arg_bits));
		gyo_hist_serial *s = &stats[(size_t)keyidx[i] * GYO_MAX_BUCKETS + b];

		s->sum += v;
		s->count++;
		total[keyidx[i]]++;
		if (maxv[keyidx[i]] < v) maxv[keyidx[i]] = v;
	}
}

/* CONN_BITMAP::add_response / get_conn_breakup common/gy_socket_stat.h:403-431 */
void gyo_conn_bitmap_add(uint16_t respmap[32], uint16_t cli_port, uint8_t bucket) { respmap[cli_port & 0x1F] |= (uint16_t)(1u << bucket); }

/* the listener's number per bucket: nactive_conn_arr_[r] = ipv4_conn[r] + ipv6_conn[r] (common/gy_socket_stat.cc:4141-4149) over
 * resp_bitmap_v4_ (rows 0..31 here) and resp_bitmap_v6_ (rows 32..63) */
void gyo_conn_bitmap_breakup2(const uint16_t respmap[64], uint8_t nconn_arr[15])
{
	uint8_t a[15], b[15];

	gyo_conn_bitmap_breakup(respmap, a);
	gyo_conn_bitmap_breakup(respmap + 32, b);
	for (int j = 0; j < 15; j++) nconn_arr[j] = (uint8_t)(a[j] + b[j]);
}

void gyo_conn_bitmap_breakup(const uint16_t respmap[32], uint8_t nconn_arr[15])
{
	for (int j = 0; j < 15; j++) {
		uint8_t n = 0;
		for (int i = 0; i < 32; i++) n += (respmap[i] >> j) & 1;
		nconn_arr[j] = n;
	}
}

/* ================================================================ HLL / CMS (builder-defined; DESIGN.md "sketch definitions") */

/* idx = top p bits; rank = 1 + number of leading zeros of the remaining (64-p) bits, capped at 64-p+1 */
void gyo_hll_idx_rank(uint64_t h64, int p, uint32_t *idx, uint8_t *rank)
{
	const uint64_t w = h64 << p;
	uint8_t r;

	*idx = (uint32_t)(h64 >> (64 - p));
	if (w == 0) {
		r = (uint8_t)(64 - p + 1);
	} else {
		r = (uint8_t)(__builtin_clzll(w) + 1);
	}
	*rank = r;
}

void gyo_hll_add(uint8_t *regs, int p, uint64_t h64)
{
	uint32_t idx;
	uint8_t r;
	gyo_hll_idx_rank(h64, p, &idx, &r);
	if (regs[idx] < r) regs[idx] = r;
}

void gyo_hll_add_words(uint8_t *regs, int p, const uint32_t *words, uint32_t nwords) { gyo_hll_add(regs, p, gyo_hash64(words, nwords)); }

void gyo_hll_merge(uint8_t *dst, const uint8_t *src, int p)
{
	for (uint32_t i = 0; i < (1u << p); i++)
		if (dst[i] < src[i]) dst[i] = src[i];
}

/* Flajolet et al. raw estimator with linear-counting small-range correction (64-bit hash: no large-range correction) */
double gyo_hll_estimate(const uint8_t *regs, int p)
{
	const uint32_t m = 1u << p;
	double sum = 0.0, alpha, e;
	uint32_t zeros = 0;

	for (uint32_t i = 0; i < m; i++) {
		sum += 1.0 / (double)(1ull << regs[i]);
		zeros += regs[i] == 0;
	}
	alpha = m == 16 ? 0.673 : (m == 32 ? 0.697 : (m == 64 ? 0.709 : 0.7213 / (1.0 + 1.079 / (double)m)));
	e = alpha * (double)m * (double)m / sum;
	if (e <= 2.5 * (double)m && zeros) {
		/* m * ln(m / zeros) without libm: series-free log via frexp-style reduction is overkill for test infra; use
		 * the builtin (gcc folds to libm call; link with -lm) */
		e = (double)m * __builtin_log((double)m / (double)zeros);
	}
	return e;
}

/* row r column = jhash2(key, n, 0xceedfead + r) & (W-1) */
void gyo_cms_cols(const uint32_t *words, uint32_t nwords, uint32_t cols[GYO_CMS_D])
{
	for (uint32_t r = 0; r < GYO_CMS_D; r++) cols[r] = gyo_jhash2(words, nwords, GYO_SEED + r) & (GYO_CMS_W - 1);
}

void gyo_cms_add(uint32_t *tbl, const uint32_t *words, uint32_t nwords, uint32_t weight)
{
	uint32_t cols[GYO_CMS_D];
	gyo_cms_cols(words, nwords, cols);
	for (uint32_t r = 0; r < GYO_CMS_D; r++) tbl[(size_t)r * GYO_CMS_W + cols[r]] += weight;
}

uint32_t gyo_cms_query(const uint32_t *tbl, const uint32_t *words, uint32_t nwords)
{
	uint32_t cols[GYO_CMS_D], m = 0xFFFFFFFFu;
	gyo_cms_cols(words, nwords, cols);
	for (uint32_t r = 0; r < GYO_CMS_D; r++) {
		uint32_t v = tbl[(size_t)r * GYO_CMS_W + cols[r]];
		if (v < m) m = v;
	}
	return m;
}

void gyo_cms64_add(uint64_t *tbl, const uint32_t *words, uint32_t nwords, uint64_t weight)
{
	uint32_t cols[GYO_CMS_D];
	gyo_cms_cols(words, nwords, cols);
	for (uint32_t r = 0; r < GYO_CMS_D; r++) tbl[(size_t)r * GYO_CMS_W + cols[r]] += weight;
}

uint64_t gyo_cms64_query(const uint64_t *tbl, const uint32_t *words, uint32_t nwords)
{
	uint32_t cols[GYO_CMS_D];
	uint64_t m = ~0ull;
	gyo_cms_cols(words, nwords, cols);
	for (uint32_t r = 0; r < GYO_CMS_D; r++) {
		uint64_t v = tbl[(size_t)r * GYO_CMS_W + cols[r]];
		if (v < m) m = v;
	}
	return m;
}

/* ================================================================ t-digest: k-bucketed merging digest in exact integers
 *
 * delta = 100 (the compression the reference passes to the Postgres tdigest extension, common/gy_query_common.cc:1855).
 * Cluster j covers the quantile range [BND[j], BND[j+1]) / 2^32 with BND[j] = round(2^32 * (1 + sin(pi*(j/100 - 1/2)))/2):
 * the k1 scale function k(q) = delta/(2 pi) asin(2q-1) cut at every half unit of k.  An item (old cluster or new value) whose
 * weighted mid-point in the merged order is mid2/2 out of N goes to cluster max{j : BND[j]*2N <= mid2*2^32}.
 * Merged order: by mean (exact rational compare); equal means: old clusters (by index) before new values.
 * All arithmetic is integer => result depends only on (old digest, multiset of new values).
 */
#include "../include/gys_tdigest_tbl.h" /* shared constant table (generated by tools/gen_tdigest_tbl.py): data, not code */

const uint64_t gyo_td_bnd[GYO_TD_NB + 1] = GYS_TDIGEST_BND_INIT;

void gyo_td_init(gyo_tdigest *d)
{
	memset(d, 0, sizeof(*d));
	d->vmin = INT_MAX;
	d->vmax = INT_MIN;
}

uint64_t gyo_td_total(const gyo_tdigest *d)
{
	uint64_t n = 0;
	for (int i = 0; i < GYO_TD_NB; i++) n += d->cnt[i];
	return n;
}

uint32_t gyo_td_cluster(uint64_t mid2, uint64_t twoN)
{
	const unsigned __int128 rhs = (unsigned __int128)mid2 << 32;
	uint32_t lo = 0, hi = GYO_TD_NB - 1; /* answer in [lo, hi]; BND[0] = 0 always satisfies */

	while (lo < hi) {
		const uint32_t mid = (lo + hi + 1) / 2;
		if ((unsigned __int128)gyo_td_bnd[mid] * twoN <= rhs)
			lo = mid;
		else
			hi = mid - 1;
	}
	return lo;
}

static int cmp_i32(const void *a, const void *b)
{
	const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
	return (x > y) - (x < y);
}

typedef struct {
	int64_t sum;
	uint64_t cnt;
} td_item;

/* generic: merge sorted weighted items (new) into d.  Items with equal mean to an old cluster go after it. */
static void td_merge_items(gyo_tdigest *d, const td_item *items, size_t m)
{
	gyo_tdigest out;
	uint64_t nold = gyo_td_total(d), nnew = 0, twoN;
	uint64_t wold_before = 0;

	for (size_t i = 0; i < m; i++) nnew += items[i].cnt;
	if (nnew == 0) return;
	twoN = 2 * (nold + nnew);
	gyo_td_init(&out);
	out.vmin = d->vmin;
	out.vmax = d->vmax;

	/* old clusters: W = (old weight before j) + (new weight with mean strictly < mean_j) */
	{
		size_t p = 0;
		uint64_t new_lt = 0;
		for (int j = 0; j < GYO_TD_NB; j++) {
			if (!d->cnt[j]) continue;
			/* advance p over new items with mean < mean_j :  s_i/c_i < S/C  <=>  s_i*C < S*c_i */
			while (p < m && (__int128)items[p].sum * (__int128)d->cnt[j] < (__int128)d->sum[j] * (__int128)items[p].cnt) {
				new_lt += items[p].cnt;
				p++;
			}
			{
				const uint64_t mid2 = 2 * (wold_before + new_lt) + d->cnt[j];
				const uint32_t c = gyo_td_cluster(mid2, twoN);
				out.sum[c] += d->sum[j];
				out.cnt[c] += d->cnt[j];
			}
			wold_before += d->cnt[j];
		}
	}
	/* new items: W = (new weight before i) + (old weight with mean <= mean_i) */
	{
		int j = 0;
		uint64_t old_le = 0, new_before = 0;
		for (size_t i = 0; i < m; i++) {
			while (j < GYO_TD_NB) {
				if (!d->cnt[j]) {
					j++;
					continue;
				}
				/* mean_j <= mean_i  <=>  S*c_i <= s_i*C */
				if ((__int128)d->sum[j] * (__int128)items[i].cnt <= (__int128)items[i].sum * (__int128)d->cnt[j]) {
					old_le += d->cnt[j];
					j++;
				} else
					break;
			}
			{
				const uint64_t mid2 = 2 * (new_before + old_le) + items[i].cnt;
				const uint32_t c = gyo_td_cluster(mid2, twoN);
				out.sum[c] += items[i].sum;
				out.cnt[c] += (uint32_t)items[i].cnt;
			}
			new_before += items[i].cnt;
		}
	}
	*d = out;
}

void gyo_td_merge_values(gyo_tdigest *d, const int32_t *vals, size_t m)
{
	int32_t *s;
	td_item *it;

	if (!m) return;
	s = (int32_t *)malloc(m * sizeof(int32_t));
	it = (td_item *)malloc(m * sizeof(td_item));
	memcpy(s, vals, m * sizeof(int32_t));
	qsort(s, m, sizeof(int32_t), cmp_i32);
	for (size_t i = 0; i < m; i++) {
		it[i].sum = s[i];
		it[i].cnt = 1;
	}
	{
		const int32_t mn = s[0], mx = s[m - 1];
		td_merge_items(d, it, m);
		if (mn < d->vmin) d->vmin = mn;
		if (mx > d->vmax) d->vmax = mx;
	}
	free(s);
	free(it);
}

void gyo_td_merge_digest(gyo_tdigest *d, const gyo_tdigest *o)
{
	td_item it[GYO_TD_NB];
	size_t m = 0;

	for (int j = 0; j < GYO_TD_NB; j++) {
		if (o->cnt[j]) {
			it[m].sum = o->sum[j];
			it[m].cnt = o->cnt[j];
			m++;
		}
	}
	if (!m) return;
	td_merge_items(d, it, m);
	if (o->vmin < d->vmin) d->vmin = o->vmin;
	if (o->vmax > d->vmax) d->vmax = o->vmax;
}

/* Quantile by linear interpolation between cluster centres (centre of cluster k at cumulative weight W_{k-1} + cnt_k/2),
 * clamped to [vmin, vmax] at the ends.  Only + - * / on doubles (IEEE exact, no contraction) so CPU and GPU agree bit-for-bit. */
static double td_quantile_interp(const gyo_tdigest *d, double q)
{
	const uint64_t N = gyo_td_total(d);
	double t, wbefore = 0.0, prev_c = 0.0, prev_mean = 0.0;
	int have_prev = 0;

	if (!N) return 0.0;
	if (q < 0.0) q = 0.0;
	if (q > 1.0) q = 1.0;
	t = q * (double)N;
	for (int k = 0; k < GYO_TD_NB; k++) {
		double c, mean;
		if (!d->cnt[k]) continue;
		mean = (double)d->sum[k] / (double)d->cnt[k];
		c = wbefore + (double)d->cnt[k] * 0.5;
		if (t < c) {
			if (!have_prev) {
				/* left tail: between vmin (at weight 0) and the first centre */
				const double lo = (double)d->vmin;
				if (c <= 0.0) return mean;
				return lo + (mean - lo) * (t / c);
			}
			return prev_mean + (mean - prev_mean) * ((t - prev_c) / (c - prev_c));
		}
		wbefore += (double)d->cnt[k];
		prev_c = c;
		prev_mean = mean;
		have_prev = 1;
	}
	{
		/* right tail: between the last centre and vmax (at weight N) */
		const double hi = (double)d->vmax, span = (double)N - prev_c;
		if (span <= 0.0) return hi;
		return prev_mean + (hi - prev_mean) * ((t - prev_c) / span);
	}
}

/* The values are integer milliseconds (tresp_msec, gy_socket_stat.cc:1519): the reported quantile is the interpolated value rounded
 * half-up to the value domain, so that it can be ranked against the data without landing between two neighbouring integers. */
double gyo_td_quantile(const gyo_tdigest *d, double q) { return floor(td_quantile_interp(d, q) + 0.5); }

/* ---- buffered form (see gy_oracle.h) */
#if GYO_TD_PEND_CAP != GYS_TDIGEST_PEND_CAP
#error "GYO_TD_PEND_CAP must equal the shared constant GYS_TDIGEST_PEND_CAP"
#endif

void gyo_tdb_init(gyo_td_buffered *b)
{
	memset(b, 0, sizeof(*b));
	gyo_td_init(&b->d);
}

void gyo_tdb_init_cap(gyo_td_buffered *b, uint32_t cap)
{
	gyo_tdb_init(b);
	b->cap = cap;
	if (cap > GYO_TD_PEND_CAP) b->ext = (int32_t *)malloc((size_t)cap * sizeof(int32_t));
}

void gyo_tdb_free(gyo_td_buffered *b)
{
	free(b->ext);
	b->ext = NULL;
}

static uint32_t tdb_cap(const gyo_td_buffered *b) { return b->cap ? b->cap : GYO_TD_PEND_CAP; }
static int32_t *tdb_buf(gyo_td_buffered *b) { return b->ext ? b->ext : b->pend; }
const int32_t *gyo_tdb_values(const gyo_td_buffered *b) { return b->ext ? b->ext : b->pend; }

uint64_t gyo_tdb_total(const gyo_td_buffered *b) { return gyo_td_total(&b->d) + b->npend; }

void gyo_tdb_add_batch(gyo_td_buffered *b, const int32_t *vals, size_t m)
{
	/* the engine's fast merge class for this buffer size: the smallest of 1024 / 2048 / 4096 values that leaves 128 of room (896 -> 1024 =
	 * GYS_TDIGEST_MERGE_FAST) */
	const size_t cap = tdb_cap(b), fast = cap + 128 <= 1024 ? 1024 : cap + 128 <= 2048 ? 2048 : 4096;
	int32_t *buf = tdb_buf(b);

	if (!m) return;
	if ((size_t)b->npend + m <= cap) {
		for (size_t i = 0; i < m; i++) {
			buf[b->npend + i] = vals[i];
			if (vals[i] < b->d.vmin) b->d.vmin = vals[i];
			if (vals[i] > b->d.vmax) b->d.vmax = vals[i];
		}
		b->npend += (uint32_t)m;
		/* one more batch like this one would take the buffer past the size the engine re-clusters fastest: re-cluster now (the
		 * per-key rate decides how full a buffer gets; without this rule a key with 200 - 500 values per batch would always merge
		 * just above that size) */
		if ((size_t)b->npend + m > fast) {
			gyo_td_merge_values(&b->d, buf, b->npend);
			b->npend = 0;
		}
	} else {
		int32_t *all = (int32_t *)malloc(((size_t)b->npend + m) * sizeof(int32_t));
		memcpy(all, buf, (size_t)b->npend * sizeof(int32_t));
		memcpy(all + b->npend, vals, m * sizeof(int32_t));
		gyo_td_merge_values(&b->d, all, (size_t)b->npend + m);
		b->npend = 0;
		free(all);
	}
}

void gyo_tdb_merged_view(const gyo_td_buffered *b, gyo_tdigest *out)
{
	*out = b->d;
	if (b->npend) gyo_td_merge_values(out, gyo_tdb_values(b), b->npend);
}

double gyo_tdb_quantile(const gyo_td_buffered *b, double q)
{
	gyo_tdigest t;
	gyo_tdb_merged_view(b, &t);
	return gyo_td_quantile(&t, q);
}

/* ================================================================ wire records + roll-ups */

static uint32_t rd_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd_u64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint16_t rd_u16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }

/* LISTENER_STATE_NOTIFY::get_elem_size common/gy_comm_proto.h:2229-2232 (+ get_act_size :2249): 88 + issue_string_len_ + padding_len_ */
uint32_t gyo_listener_state_elem_size(const uint8_t *rec) { return GYO_LISTENER_STATE_NOTIFY_SZ + rec[85] + rec[86]; }

/* TCP_CONN_NOTIFY::get_elem_size common/gy_comm_proto.h:1721-1724: 280 + cli_cmdline_len_ + padding_len_ */
uint32_t gyo_tcp_conn_elem_size(const uint8_t *rec) { return GYO_TCP_CONN_NOTIFY_SZ + rd_u16(rec + 272) + rec[279]; }

/* ---------------------------------------------------------------- wire framing: L1 validation of a partha -> madhava message
 * COMM_HEADER (16 B: magic_, total_sz_, data_type_, padding_sz_) + EVENT_NOTIFY (8 B: subtype_, nevents_) + records
 * (common/gy_comm_proto.h:336-420, :486-500).  PINNED: tests/test_wire.py runs these against the reference's own validators
 * (common/gy_comm_proto.cc compiled into oracle/_ref). */
#define GYO_PM_HDR_MAGIC 0x05666605u
#define GYO_COMM_EVENT_NOTIFY 14u
#define GYO_COMM_MIN_TYPE 1u
#define GYO_COMM_MAX_TYPE 18u
#define GYO_MAX_COMM_DATA_SZ (16u << 20)

/* COMM_HEADER::validate (common/gy_comm_proto.cc:10-57) for the message types the engine looks into (EVENT_NOTIFY); any other
 * in-range data_type_ is only checked for the generic header rules (the engine skips those messages) */
int gyo_comm_header_validate(const uint8_t *msg, uint32_t req_magic)
{
	const uint32_t magic = rd_u32(msg), total_sz = rd_u32(msg + 4), data_type = rd_u32(msg + 8), padding_sz = rd_u32(msg + 12);
	if (!(magic == req_magic && total_sz < GYO_MAX_COMM_DATA_SZ && total_sz >= 16 && padding_sz < 8 && data_type > GYO_COMM_MIN_TYPE &&
	      data_type < GYO_COMM_MAX_TYPE))
		return 0;
	if (total_sz & 7u) return 0;
	if (((uintptr_t)msg) & 7u) return 0; /* "We will terminate connections resulting in unaligned accesses" */
	if (data_type == GYO_COMM_EVENT_NOTIFY) return (total_sz - padding_sz) >= 16 + 8;
	return 1;
}

/* TCP_CONN_NOTIFY::validate (:840-881) / LISTENER_STATE_NOTIFY::validate (:955-996): nevents_ records of get_elem_size() bytes,
 * each a multiple of 8, inside the message's actual length */
static int chain_validate(const uint8_t *msg, uint32_t fixed, uint32_t max_elems, uint32_t (*elem_size)(const uint8_t *))
{
	const int64_t act = (int64_t)rd_u32(msg + 4) - (int64_t)rd_u32(msg + 12);
	if (act < 16 + 8) return 0;
	const uint32_t nelems = rd_u32(msg + 20);
	if (nelems > max_elems) return 0;
	int64_t totallen = act - (16 + 8);
	const uint8_t *p = msg + 24;
	uint32_t i;
	for (i = 0; i < nelems && totallen >= (int64_t)fixed; ++i) {
		const int64_t sz = (int64_t)elem_size(p);
		if (totallen < sz) return 0;
		if (sz & 7) return 0; /* "Padding issue" */
		totallen -= sz;
		p += sz;
	}
	return i == nelems;
}

int gyo_tcp_conn_validate(const uint8_t *msg) { return chain_validate(msg, GYO_TCP_CONN_NOTIFY_SZ, 2048u, gyo_tcp_conn_elem_size); }
int gyo_listener_state_validate(const uint8_t *msg) { return chain_validate(msg, GYO_LISTENER_STATE_NOTIFY_SZ, 512u, gyo_listener_state_elem_size); }

/* partha_listener_state loop server/gy_mconnhdlr.cc:11175-11256 + LISTEN_SUMM_STATS::update server/gy_msocket.h:853-865.
 * Records flagged LISTEN_FLAG_DELETE (0xC0, gy_comm_proto.h:2180) are skipped before the update (:11194-11248). */
int gyo_listener_state_rollup(const uint8_t *batch, int nrec, const uint8_t *pend, gyo_listen_summ_stats *summ, int *nerrors)
{
	const uint8_t *p = batch;
	int i;

	for (i = 0; i < nrec && p < pend; ++i, p += gyo_listener_state_elem_size(p)) {
		const uint8_t curr_state = p[79], query_flags = p[84];
		const uint32_t nqrys_5s = rd_u32(p + 8);

		if (query_flags == 0xC0) continue;
		if (curr_state > 5) {
			if (nerrors) (*nerrors)++;
			continue;
		}
		summ->nstates[curr_state]++;
		summ->tot_qps += (int32_t)(nqrys_5s / 5);
		summ->tot_act_conn += (int32_t)rd_u32(p + 20);
		summ->tot_kb_inbound += (int32_t)rd_u32(p + 36);
		summ->tot_kb_outbound += (int32_t)rd_u32(p + 40);
		summ->tot_ser_errors += (int32_t)rd_u32(p + 44);
		summ->nlisteners++;
		summ->nactive += !!nqrys_5s;
	}
	return i;
}

/* IP_PORT object at rec+off (32 bytes: ip128 @0, ip32 @16, aftype @20, flags @22, port @24) as the sender's GY_IP_ADDR stands in the record: the
 * receiver hashes the object it was sent (get_as_inaddr reads the ip32_be_ / ip128_be_ members, it does not rebuild them) */
static void rd_ip_port(const uint8_t *p, const uint8_t **ip128, uint32_t *ip32, uint16_t *port)
{
	*ip32 = rd_u32(p + 16);
	*ip128 = p;
	*port = rd_u16(p + 24);
}

/* partha_tcp_conn_info walk server/gy_mconnhdlr.cc:9130 ; flow key PAIR_IP_PORT(nat_cli_, nat_ser_) :8707 */
int gyo_tcp_conn_decode(const uint8_t *batch, int nrec, const uint8_t *pend, uint32_t *keywords, uint32_t *nwords,
			uint64_t *ser_glob_id, uint64_t *bytes_sent, uint64_t *bytes_rcvd, uint8_t *flags)
{
	const uint8_t *p = batch;
	int i;

	for (i = 0; i < nrec && p < pend; ++i, p += gyo_tcp_conn_elem_size(p)) {
		const uint8_t *cip, *sip;
		uint32_t c32, s32;
		uint16_t cport, sport;

		rd_ip_port(p + 64, &cip, &c32, &cport);
		rd_ip_port(p + 96, &sip, &s32, &sport);
		nwords[i] = gyo_pair_words_obj(c32, cip, cport, s32, sip, sport, keywords + (size_t)i * 10);
		ser_glob_id[i] = rd_u64(p + 192);
		bytes_sent[i] = rd_u64(p + 208);
		bytes_rcvd[i] = rd_u64(p + 216);
		if (flags) flags[i] = (uint8_t)((p[274] ? 1 : 0) | (p[275] ? 2 : 0) | (p[276] ? 4 : 0) | (p[277] ? 8 : 0) | (p[278] ? 16 : 0));
	}
	return i;
}

/* The five bool bytes of a TCP_CONN_NOTIFY record (common/gy_comm_proto.h:1700-1704, @274..278) and what the walk of
 * MCONN_HANDLER::partha_tcp_conn_info makes of them (server/gy_mconnhdlr.cc:9129-9341):
 *   conn_closed = !!tusec_close_ (:9131); a closed record counts nclosed, and nclosed_no_not when notified_before_ is clear (:9133-9137);
 *   an open record counts nnew (:9327) and goes to add_tcp_conn_cli when is_tcp_connect_event_, else to add_tcp_conn_ser (:9329-9339);
 *   the per-listener close roll-up connlistenmap_ takes closed records with bytes whose ser_glob_id_ is set and is_tcp_accept_event_
 *   (:9226-9245), the per-client one connclientmap_ those that are connect-only with cli_task_aggr_id_ set (:9290-9312).
 * A connection is thus reported once when it opens and once when it closes (or once only, at its close, when it was short-lived), by each
 * of its two halves; is_loopback_conn_ = both flags on the one record of a same-host connection (common/gy_socket_stat.cc:1738-1787). */
static int conn_closed(const uint8_t *p) { return rd_u64(p + 136) != 0; }
static int conn_fresh(const uint8_t *p) { return p[278] == 0; }               /* notified_before_ clear */
static int conn_listener_side(const uint8_t *p) { return p[275] != 0 || p[274] == 0; } /* accept event, or not a connect event (:9333) */

/* whole-batch form of the TCP_CONN_NOTIFY roll-up used by the engine: walk (:9130), flow key into the distinct-flow HLL (every record:
 * both halves and both notifications of a connection carry the same NAT-translated tuple), service key (ser_glob_id_) into the Count-Min
 * rows ONCE PER CONNECTION: the listener-side record whose notified_before_ is clear adds the connection, listener-side records add their
 * bytes_sent_ + bytes_rcvd_ (zero while a connection is open).  Returns the records walked. */
int gyo_tcp_conn_sketch_batch(const uint8_t *batch, int nrec, const uint8_t *pend, uint8_t *hll, uint32_t *cms32, uint64_t *cms64)
{
	const uint8_t *p = batch;
	int i;

	for (i = 0; i < nrec && p < pend; ++i, p += gyo_tcp_conn_elem_size(p)) {
		const uint8_t *cip, *sip;
		uint32_t c32, s32;
		uint16_t cport, sport;
		uint32_t w[10], gw[2], nw;
		uint64_t gid, bytes;

		rd_ip_port(p + 64, &cip, &c32, &cport);
		rd_ip_port(p + 96, &sip, &s32, &sport);
		nw = gyo_pair_words_obj(c32, cip, cport, s32, sip, sport, w);
		gyo_hll_add_words(hll, GYO_HLL_P, w, nw);
		if (!conn_listener_side(p)) continue;
		gid = rd_u64(p + 192);
		gw[0] = (uint32_t)(gid & 0xFFFFFFFFu);
		gw[1] = (uint32_t)(gid >> 32);
		if (conn_fresh(p)) gyo_cms_add(cms32, gw, 2, 1);
		bytes = rd_u64(p + 208) + rd_u64(p + 216);
		if (bytes) gyo_cms64_add(cms64, gw, 2, bytes);
	}
	return i;
}

/* the walk's own tallies (its DEBUG line :9431-9437): out[0] nnew, out[1] nclosed, out[2] nclosed_no_not, out[3] records of the
 * connecting half only (added to, not cleared) */
int gyo_tcp_conn_walk_tallies(const uint8_t *batch, int nrec, const uint8_t *pend, uint64_t out[4])
{
	const uint8_t *p = batch;
	int i;

	for (i = 0; i < nrec && p < pend; ++i, p += gyo_tcp_conn_elem_size(p)) {
		if (conn_closed(p)) {
			out[1]++;
			if (conn_fresh(p)) out[2]++;
		} else
			out[0]++;
		if (!conn_listener_side(p)) out[3]++;
	}
	return i;
}

/* per-service connection counters the engine exports (gys_export_svc_counters): ctr[4 s + {0,1,2,3}] = connections, closes, bytes sent,
 * bytes received of service gids[s], from the listener-side records only; *unknown += listener-side records of services not in gids */
int gyo_tcp_conn_svc_counters(const uint8_t *batch, int nrec, const uint8_t *pend, const uint64_t *gids, uint32_t ngids, uint64_t *ctr, uint64_t *unknown)
{
	const uint8_t *p = batch;
	int i;

	for (i = 0; i < nrec && p < pend; ++i, p += gyo_tcp_conn_elem_size(p)) {
		const uint64_t gid = rd_u64(p + 192);
		uint32_t s;

		if (!conn_listener_side(p)) continue;
		for (s = 0; s < ngids && gids[s] != gid; ++s)
			;
		if (s == ngids) {
			(*unknown)++;
			continue;
		}
		ctr[4 * s + 0] += conn_fresh(p);
		ctr[4 * s + 1] += conn_closed(p);
		ctr[4 * s + 2] += rd_u64(p + 208);
		ctr[4 * s + 3] += rd_u64(p + 216);
	}
	return i;
}

/* CLUSTER_STATE_ONE::update_from_state server/gy_mconnhdlr.cc:16032-16050 */
void gyo_cluster_state_update(gyo_cluster_state_one *c, uint32_t ntasks_issue, uint32_t ntasks, uint32_t nlisten_issue,
			      uint32_t nlisten, uint32_t cpu_issue, uint32_t mem_issue, const gyo_listen_summ_stats *summ)
{
	c->nhosts++;
	c->ntasks_issue += ntasks_issue;
	c->ntaskissue_hosts += !!ntasks_issue;
	c->ntasks += ntasks;
	c->nsvc_issue += nlisten_issue;
	c->nsvcissue_hosts += !!nlisten_issue;
	c->nsvc += nlisten;
	c->total_qps += (uint32_t)summ->tot_qps;
	c->svc_net_mb += (uint32_t)((summ->tot_kb_inbound + summ->tot_kb_outbound) / 1024); /* int arithmetic then u32 += */
	c->ncpu_issue += !!cpu_issue;
	c->nmem_issue += !!mem_issue;
}

/* MS_CLUSTER_STATE::STATE_ONE::add_stats common/gy_comm_proto.h:3200-3215 */
void gyo_cluster_state_add(gyo_cluster_state_one *d, const gyo_cluster_state_one *s)
{
	d->nhosts += s->nhosts;
	d->ntasks_issue += s->ntasks_issue;
	d->ntaskissue_hosts += s->ntaskissue_hosts;
	d->ntasks += s->ntasks;
	d->nsvc_issue += s->nsvc_issue;
	d->nsvcissue_hosts += s->nsvcissue_hosts;
	d->nsvc += s->nsvc;
	d->total_qps += s->total_qps;
	d->svc_net_mb += s->svc_net_mb;
	d->ncpu_issue += s->ncpu_issue;
	d->nmem_issue += s->nmem_issue;
}

/* BOUNDED_PRIO_QUEUE::push_locked common/gy_statistics.h:356-383 with Comp = greater: a min-heap of the N largest.  The retained
 * multiset is the N largest values (a value equal to the current minimum does not displace it). */
static int cmp_u64_desc(const void *a, const void *b)
{
	const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
	return (x < y) - (x > y);
}

size_t gyo_topn_u64(const uint64_t *vals, size_t n, size_t maxn, uint64_t *out)
{
	size_t sz = 0;

	for (size_t i = 0; i < n; i++) {
		if (sz < maxn) {
			out[sz++] = vals[i];
			continue;
		}
		{
			size_t mi = 0;
			for (size_t k = 1; k < sz; k++)
				if (out[k] < out[mi]) mi = k;
			if (vals[i] > out[mi]) out[mi] = vals[i];
		}
	}
	qsort(out, sz, sizeof(uint64_t), cmp_u64_desc);
	return sz;
}
