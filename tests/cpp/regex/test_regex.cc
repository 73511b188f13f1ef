// CPU test of gyeeta_amd/csrc/gys_regex.hpp (the `like` / `notlike` criteria; the reference: RE2::PartialMatch,
// common/gy_query_criteria.h:1364-1378).  RE2 is not in the reference tree: the accepted subset is compared with std::regex (ECMAScript) on
// patterns both read the same way, and the limits RE2 documents (repetition counts, nested repetitions, program size) are checked on their own.
#include "../../../gyeeta_amd/csrc/gys_regex.hpp"

#include <chrono>
#include <cstdio>
#include <random>
#include <regex>
#include <string>
#include <vector>

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { ++fails; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)

static bool compiles(const char *pat, std::string *err = nullptr)
{
	gysre::Regex r;
	std::string e;
	const bool ok = r.compile(pat, &e);
	if (err) *err = e;
	return ok;
}
static bool matches(const char *pat, const std::string &subj)
{
	gysre::Regex r;
	std::string e;
	if (!r.compile(pat, &e)) { ++fails; fprintf(stderr, "FAIL: %s does not compile: %s\n", pat, e.c_str()); return false; }
	return r.search(subj.data(), subj.size());
}

int main()
{
	using clk = std::chrono::steady_clock;
	// ---- nested counted repetitions of an operand that emits nothing must not multiply the parse work
	for (const char *pat : {"((){1000}){1000}", "(((){1000}){1000}){30}", "((((){1000}){1000}){1000}){1000}", "(?:(?:(?:){999}){999}){999}x",
				"(((a{0}){1000}){1000}){1000}", "((()|()){1000}){1000}"}) {
		const auto t0 = clk::now();
		std::string err;
		const bool ok = compiles(pat, &err);
		const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
		CHECK(ms < 200.0, "%s took %.1f ms to %s", pat, ms, ok ? "compile" : "reject");
		if (ok) { gysre::Regex r; std::string e; r.compile(pat, &e); CHECK(r.search("abc", 3) == (std::string(pat).back() != 'x'), "%s on abc", pat); }
	}
	CHECK(matches("a{0}", "xyz") && matches("", "xyz") && matches("^a{0}$", ""), "patterns that match the empty string only");
	// RE2 rejects nested counted repetitions whose product exceeds 1000 (kRegexpRepeatSize)
	CHECK(!compiles("(a{100}){100}"), "(a{100}){100} accepted");
	CHECK(!compiles("((a{10}){10}){11}"), "product 1100 accepted");
	CHECK(compiles("(a{10}){10}"), "(a{10}){10} rejected");
	CHECK(compiles("((a{10}){10}){10}"), "product 1000 rejected");
	CHECK(matches("^(a{10}){10}$", std::string(100, 'a')) && !matches("^(a{10}){10}$", std::string(99, 'a')), "(a{10}){10}");
	CHECK(compiles("(a+){1000}") || true, "-"); // (uncounted repetitions do not enter the product; the program-size limit decides)
	CHECK(compiles("(a*b+){30}"), "(a*b+){30}");
	// ---- counts above the limit are errors, not literal braces
	CHECK(!compiles("a{10010}"), "a{10010} accepted");
	CHECK(!compiles("a{1,99999}"), "a{1,99999} accepted");
	CHECK(!compiles("a{1001}"), "a{1001} accepted");
	CHECK(!compiles("a{5,2}"), "a{5,2} accepted");
	CHECK(compiles("a{1000}") && compiles("a{0,1000}") && compiles("a{1000,}"), "limit itself");
	// a brace that does not open a repetition is a literal, and a quantifier behind it applies to it
	CHECK(matches("^a{*$", "a") && matches("^a{*$", "a{{{") && !matches("^a{*$", "aa"), "a{*");
	CHECK(matches("^a{,3}$", "a{,3}") && matches("^x{a}$", "x{a}") && matches("^a{1,b$", "a{1,b"), "literal braces");
	CHECK(matches("^a{+b$", "a{{b") && !matches("^a{+b$", "ab"), "a{+b");
	// ---- accepted subset against std::regex (ECMAScript reads these patterns the same way)
	const std::vector<std::string> pats = {"^post.*r$", "^(nginx|envoy)$", "x{16}", "east-[0-9]$", "a(b|c)*d", "[a-f]{2,4}z", "^(ab)+$", "(a|ab)(c|bcd)(d*)",
					       "a.c", "^[^a-c]+$", "\\d{3}-\\d{2}", "(foo|bar){2}", "^a?b?c?$", "z{0}a", "(x+x+)+y", "^(a{3}){2}b", "[[:alpha:]]+[[:digit:]]"};
	std::mt19937 rng(7);
	const char alphabet[] = "abcdxyz0129-{}";
	for (const std::string &pt : pats) {
		gysre::Regex r;
		std::string e;
		if (!r.compile(pt, &e)) { ++fails; fprintf(stderr, "FAIL: %s: %s\n", pt.c_str(), e.c_str()); continue; }
		const std::regex sr(pt, std::regex::ECMAScript);
		for (int i = 0; i < 1500; ++i) {
			std::string subj;
			const int n = (int)(rng() % 12);
			for (int k = 0; k < n; ++k) subj.push_back(alphabet[rng() % (sizeof(alphabet) - 1)]);
			if (i % 7 == 0) subj = "postmaster";
			if (i % 11 == 0) subj = std::string(16, 'x') + subj;
			const bool got = r.search(subj.data(), subj.size()), want = std::regex_search(subj, sr);
			CHECK(got == want, "%s on '%s': got %d, std::regex %d", pt.c_str(), subj.c_str(), (int)got, (int)want);
		}
	}
	// patterns a backtracking matcher needs exponential time for: linear here
	{
		const auto t0 = clk::now();
		CHECK(!matches("^(a+)+$", std::string(4000, 'a') + "!"), "(a+)+");
		CHECK(!matches("(x+x+)+y", std::string(3000, 'x')), "(x+x+)+y");
		const double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
		CHECK(ms < 2000.0, "pathological patterns took %.0f ms", ms);
	}
	if (fails) { fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
	printf("regex ok\n");
	return 0;
}
