// gys_regex.hpp -- a small LINEAR-TIME regular-expression matcher for the `like` / `notlike` string criteria (host code only).
//
// The reference evaluates `like` with RE2::PartialMatch and set_max_mem(1 << 20) (common/gy_query_criteria.h:1364-1378, :330-360): an
// automaton, never a backtracking search -- a pattern that comes in through a web query cannot make a criterion take exponential time.
// RE2 is not part of the reference tree (and not installed here), so its published behaviour for the syntax an operator writes in such
// criteria is restated as a Thompson construction run as a Pike VM (one pass over the subject, a set of program counters per byte:
// O(len(subject) x len(program)) whatever the pattern).  std::regex, used before round 5, backtracks.
//
// Syntax (byte oriented, as RE2 in Latin-1 mode): literals; . ; [...] with ranges, negation, escapes and the POSIX classes [[:alpha:]] ...;
// \d \D \w \W \s \S; \b \B; ^ $ \A \z; \n \t \r \f \v \a \xHH \x{HH} \0 and escaped punctuation; ( ) (?: ) (?P<name> ); | ; * + ? {n} {n,}
// {n,m} and their lazy forms (the same language, and a boolean match does not see the preference); the flags i, s, m, U as (?i) / (?i: ).
// Rejected, as RE2 rejects them: back-references, look-around, possessive quantifiers, repetition counts above 1000.  Rejected here though
// RE2 takes them: Unicode classes (\pL, \p{Greek}) and \C, \Q..\E -- a pattern that needs them gets GYS_ERR_INVAL instead of a different
// meaning.  A program larger than GYS_RE_MAX_PROG instructions is rejected (RE2: "pattern too large - compile failed").
#pragma once

#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

namespace gysre {

#define GYS_RE_MAX_PATTERN 1024u
#define GYS_RE_MAX_PROG 16384u
#define GYS_RE_MAX_REPEAT 1000u
#define GYS_RE_MAX_ATOMS 262144u // parse steps of one compile, re-parsed copies of repeated operands included

struct CharSet {
	uint64_t w[4] = {0, 0, 0, 0};
	void add(unsigned c) { w[c >> 6] |= 1ull << (c & 63); }
	void add_range(unsigned a, unsigned b) { for (unsigned c = a; c <= b; ++c) add(c); }
	bool has(unsigned c) const { return (w[c >> 6] >> (c & 63)) & 1ull; }
	void negate() { for (auto &x : w) x = ~x; }
	void merge(const CharSet &o) { for (int i = 0; i < 4; ++i) w[i] |= o.w[i]; }
	void fold_case()
	{
		for (unsigned c = 'a'; c <= 'z'; ++c)
			if (has(c) || has(c - 32)) {
				add(c);
				add(c - 32);
			}
	}
};

enum Op : uint8_t { OP_CHAR, OP_SET, OP_SPLIT, OP_JMP, OP_ASSERT, OP_MATCH };
enum Assert : uint8_t { A_BOL, A_EOL, A_BOT, A_EOT, A_WORDB, A_NWORDB };

struct Inst {
	Op op;
	uint8_t arg;    // OP_CHAR: the byte; OP_ASSERT: which
	uint32_t x, y;  // OP_SPLIT: both targets; OP_JMP: x; OP_SET: x = index into sets
};

class Regex {
public:
	// returns false and sets err on a pattern that is not accepted
	bool compile(const std::string &pat, std::string *err)
	{
		prog_.clear();
		sets_.clear();
		p_ = pat.data();
		end_ = p_ + pat.size();
		err_.clear();
		rep_mult_ = 1;
		atoms_parsed_ = 0;
		if (pat.size() > GYS_RE_MAX_PATTERN) return fail(err, "pattern longer than 1024 bytes");
		Flags f;
		Frag fr;
		if (!parse_alt(f, 0, &fr) || p_ != end_) {
			if (err_.empty()) err_ = p_ != end_ ? "unmatched )" : "bad pattern";
			return fail(err, err_.c_str());
		}
		const uint32_t m = emit(Inst{OP_MATCH, 0, 0, 0});
		patch(fr, m);
		start_ = fr.empty ? m : fr.start; // (a pattern that matches only the empty string -- "", "a{0}" -- starts at the match instruction: an unreachable operand may sit at 0)
		if (prog_.size() > GYS_RE_MAX_PROG) return fail(err, "pattern too large");
		return true;
	}

	// RE2::PartialMatch: does some substring of s[0, n) match?
	bool search(const char *s, size_t n) const
	{
		const uint32_t np = (uint32_t)prog_.size();
		if (!np) return false;
		std::vector<uint32_t> dense_a(np), dense_b(np), sparse(np), stack;
		stack.reserve(64);
		uint32_t na = 0, nb = 0;
		uint32_t *cur = dense_a.data(), *nxt = dense_b.data();
		// a list is a sparse set: member(pc) <=> sparse[pc] < n && dense[sparse[pc]] == pc; each list is built once per position
		auto add_thread = [&](uint32_t *dense, uint32_t &cnt, uint32_t pc0, size_t pos) -> bool {
			stack.clear();
			stack.push_back(pc0);
			while (!stack.empty()) {
				const uint32_t pc = stack.back();
				stack.pop_back();
				if (sparse[pc] < cnt && dense[sparse[pc]] == pc) continue;
				sparse[pc] = cnt;
				dense[cnt++] = pc;
				const Inst &in = prog_[pc];
				switch (in.op) {
				case OP_JMP: stack.push_back(in.x); break;
				case OP_SPLIT:
					stack.push_back(in.y);
					stack.push_back(in.x);
					break;
				case OP_ASSERT:
					if (assert_ok((Assert)in.arg, s, n, pos)) stack.push_back(in.x);
					break;
				case OP_MATCH: return true;
				default: break;
				}
			}
			return false;
		};
		for (size_t pos = 0;; ++pos) {
			// unanchored: a new thread starts at every position (the list is a set, so this costs one probe when it is already there)
			if (add_thread(cur, na, start_, pos)) return true;
			if (pos == n) break;
			const unsigned c = (unsigned char)s[pos];
			nb = 0;
			// (the sparse array is shared by both lists: membership is decided against the list being BUILT, whose dense array is nxt)
			for (uint32_t i = 0; i < na; ++i) {
				const Inst &in = prog_[cur[i]];
				bool ok = false;
				if (in.op == OP_CHAR) ok = in.arg == c;
				else if (in.op == OP_SET) ok = sets_[in.x].has(c);
				if (ok && step_add(nxt, nb, in.op == OP_CHAR || in.op == OP_SET ? cur[i] + 1 : 0, s, n, pos + 1, sparse, stack)) return true;
			}
			std::swap(cur, nxt);
			na = nb;
		}
		return false;
	}

	size_t program_size() const { return prog_.size(); }

private:
	struct Flags {
		bool icase = false, dotnl = false, multiline = false;
	};
	struct Frag {
		uint32_t start = 0;
		std::vector<uint32_t> out; // instruction slots to patch: (pc << 1) | which (0: x, 1: y)
		bool empty = true;         // matches only the empty string and has no instruction (start is meaningless)
	};
	std::vector<Inst> prog_;
	std::vector<CharSet> sets_;
	uint32_t start_ = 0;
	const char *p_ = nullptr, *end_ = nullptr;
	std::string err_;
	uint64_t rep_mult_ = 1;     // product of the counts of the counted repetitions being expanded around the parser's position
	uint32_t atoms_parsed_ = 0; // parse_atom calls of this compile (re-parsed copies included)

	static bool fail(std::string *err, const char *msg)
	{
		if (err) *err = msg;
		return false;
	}
	bool bad(const char *msg)
	{
		if (err_.empty()) err_ = msg;
		return false;
	}
	static bool is_word(unsigned c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_'; }
	static bool assert_ok(Assert a, const char *s, size_t n, size_t pos)
	{
		switch (a) {
		case A_BOT: return pos == 0;
		case A_EOT: return pos == n;
		case A_BOL: return pos == 0 || s[pos - 1] == '\n';
		case A_EOL: return pos == n || s[pos] == '\n';
		default: {
			const bool l = pos > 0 && is_word((unsigned char)s[pos - 1]), r = pos < n && is_word((unsigned char)s[pos]);
			return (l != r) == (a == A_WORDB);
		}
		}
	}
	bool step_add(uint32_t *dense, uint32_t &cnt, uint32_t pc0, const char *s, size_t n, size_t pos, std::vector<uint32_t> &sparse, std::vector<uint32_t> &stack) const
	{
		stack.clear();
		stack.push_back(pc0);
		while (!stack.empty()) {
			const uint32_t pc = stack.back();
			stack.pop_back();
			if (sparse[pc] < cnt && dense[sparse[pc]] == pc) continue;
			sparse[pc] = cnt;
			dense[cnt++] = pc;
			const Inst &in = prog_[pc];
			switch (in.op) {
			case OP_JMP: stack.push_back(in.x); break;
			case OP_SPLIT:
				stack.push_back(in.y);
				stack.push_back(in.x);
				break;
			case OP_ASSERT:
				if (assert_ok((Assert)in.arg, s, n, pos)) stack.push_back(in.x);
				break;
			case OP_MATCH: return true;
			default: break;
			}
		}
		return false;
	}

	uint32_t emit(Inst in)
	{
		prog_.push_back(in);
		return (uint32_t)prog_.size() - 1;
	}
	void patch(const Frag &f, uint32_t target)
	{
		for (uint32_t o : f.out) {
			if (o & 1u) prog_[o >> 1].y = target;
			else prog_[o >> 1].x = target;
		}
	}
	// a fragment that consumes exactly one byte out of `cs`
	Frag frag_set(const CharSet &cs_in, const Flags &f)
	{
		CharSet cs = cs_in;
		if (f.icase) cs.fold_case();
		Frag fr;
		fr.empty = false;
		int single = -1, cnt = 0;
		for (unsigned c = 0; c < 256 && cnt < 2; ++c)
			if (cs.has(c)) {
				single = (int)c;
				++cnt;
			}
		if (cnt == 1) {
			fr.start = emit(Inst{OP_CHAR, (uint8_t)single, 0, 0});
		} else {
			sets_.push_back(cs);
			fr.start = emit(Inst{OP_SET, 0, (uint32_t)sets_.size() - 1, 0});
		}
		// a consuming instruction falls through to pc + 1: a JMP behind it is what later patches redirect
		const uint32_t j = emit(Inst{OP_JMP, 0, 0, 0});
		fr.out.push_back(j << 1);
		return fr;
	}
	Frag frag_assert(Assert a)
	{
		Frag fr;
		fr.empty = false;
		fr.start = emit(Inst{OP_ASSERT, (uint8_t)a, 0, 0});
		fr.out.push_back(fr.start << 1);
		return fr;
	}
	Frag cat(Frag a, Frag b)
	{
		if (a.empty) return b;
		if (b.empty) return a;
		patch(a, b.start);
		a.out = std::move(b.out);
		return a;
	}
	Frag alt(Frag a, Frag b)
	{
		Frag fr;
		fr.empty = false;
		fr.start = emit(Inst{OP_SPLIT, 0, 0, 0});
		if (a.empty) fr.out.push_back(fr.start << 1);
		else {
			prog_[fr.start].x = a.start;
			fr.out.insert(fr.out.end(), a.out.begin(), a.out.end());
		}
		if (b.empty) fr.out.push_back((fr.start << 1) | 1u);
		else {
			prog_[fr.start].y = b.start;
			fr.out.insert(fr.out.end(), b.out.begin(), b.out.end());
		}
		return fr;
	}
	Frag star(Frag a) // a*
	{
		if (a.empty) return a;
		Frag fr;
		fr.empty = false;
		fr.start = emit(Inst{OP_SPLIT, 0, a.start, 0});
		patch(a, fr.start);
		fr.out.push_back((fr.start << 1) | 1u);
		return fr;
	}
	Frag quest(Frag a) // a?
	{
		if (a.empty) return a;
		Frag fr;
		fr.empty = false;
		fr.start = emit(Inst{OP_SPLIT, 0, a.start, 0});
		fr.out = a.out;
		fr.out.push_back((fr.start << 1) | 1u);
		return fr;
	}

	// the parser re-parses a sub-pattern to copy it ({n,m} needs n .. m copies of the operand): an operand is remembered as its source range
	bool parse_alt(Flags f, int depth, Frag *out)
	{
		if (depth > 64) return bad("nesting too deep");
		Frag acc;
		if (!parse_cat(f, depth, &acc)) return false;
		while (p_ < end_ && *p_ == '|') {
			++p_;
			Frag rhs;
			if (!parse_cat(f, depth, &rhs)) return false;
			acc = alt(std::move(acc), std::move(rhs));
			if (prog_.size() > GYS_RE_MAX_PROG) return bad("pattern too large");
		}
		*out = std::move(acc);
		return true;
	}
	bool parse_cat(Flags &f, int depth, Frag *out)
	{
		Frag acc;
		while (p_ < end_ && *p_ != '|' && *p_ != ')') {
			const char *a0 = p_;
			Flags f_atom = f;
			Frag atom;
			bool flag_only = false;
			if (!parse_atom(f, depth, &atom, &flag_only)) return false;
			if (flag_only) continue; // (?i) and friends: the rest of this group runs under the new flags
			const char *a1 = p_;
			// quantifiers (several in a row are rejected like RE2's "bad repetition operator")
			if (p_ < end_ && (*p_ == '*' || *p_ == '+' || *p_ == '?' || *p_ == '{')) {
				uint32_t lo = 0, hi = 0; // hi = ~0: unbounded
				const uint64_t mult_outer = rep_mult_;
				bool counted = false; // an explicit {n}, {n,}, {n,m}
				if (*p_ == '*') { lo = 0; hi = ~0u; ++p_; }
				else if (*p_ == '+') { lo = 1; hi = ~0u; ++p_; }
				else if (*p_ == '?') { lo = 0; hi = 1; ++p_; }
				else {
					const char *q = p_ + 1;
					uint32_t v = 0, w = 0;
					bool have = false, comma = false, have2 = false;
					// (all digits are consumed, the value saturates just above the limit: {10010} is a bad count, not a literal brace)
					while (q < end_ && *q >= '0' && *q <= '9') { v = v > GYS_RE_MAX_REPEAT ? v : v * 10 + (uint32_t)(*q - '0'); ++q; have = true; }
					if (q < end_ && *q == ',') {
						comma = true;
						++q;
						while (q < end_ && *q >= '0' && *q <= '9') { w = w > GYS_RE_MAX_REPEAT ? w : w * 10 + (uint32_t)(*q - '0'); ++q; have2 = true; }
					}
					if (!have || q >= end_ || *q != '}') {
						// not a repetition: the '{' is a literal (RE2 does the same) -- left in place, so that it is parsed as the next atom and
						// a quantifier behind it applies to it (a{* = a, then zero or more braces)
						acc = cat(std::move(acc), std::move(atom));
						continue;
					}
					counted = true;
					lo = v;
					hi = comma ? (have2 ? w : ~0u) : v;
					if (lo > GYS_RE_MAX_REPEAT || (hi != ~0u && (hi > GYS_RE_MAX_REPEAT || hi < lo))) return bad("bad repetition count");
					p_ = q + 1;
				}
				if (p_ < end_ && *p_ == '?') ++p_; // lazy form: same language
				if (p_ < end_ && (*p_ == '*' || *p_ == '+' || *p_ == '?')) return bad("bad repetition operator");
				if (p_ < end_ && *p_ == '{') { // a{2}{3}: RE2 rejects a repetition of a repetition too
					const char *q = p_ + 1;
					if (q < end_ && *q >= '0' && *q <= '9') return bad("bad repetition operator");
				}
				// expand: lo copies, then (hi - lo) optional copies or a star.  Copies are made by re-parsing the operand's source.
				const char *resume = p_;
				if (atom.empty && atom.out.empty()) { // a repetition of nothing (an empty group) is nothing: no copies
					acc = cat(std::move(acc), std::move(atom));
					if (prog_.size() > GYS_RE_MAX_PROG) return bad("pattern too large");
					continue;
				}
				// RE2 rejects nested counted repetitions whose product exceeds 1000 (kRegexpRepeatSize, RepetitionWalker): the copies of this
				// operand are parsed under the product of the enclosing counts
				if (counted) {
					const uint64_t cnt = std::max<uint32_t>(1u, hi != ~0u ? hi : lo);
					if (rep_mult_ * cnt > GYS_RE_MAX_REPEAT) return bad("bad repetition count");
					rep_mult_ *= cnt;
				}
				auto reparse = [&](Frag *fr) -> bool {
					p_ = a0;
					Flags ff = f_atom;
					bool fo = false;
					const bool ok = parse_atom(ff, depth, fr, &fo);
					(void)a1;
					return ok;
				};
				Frag rep;
				bool first_used = false;
				auto next_copy = [&](Frag *fr) -> bool {
					if (!first_used) {
						first_used = true;
						*fr = std::move(atom);
						return true;
					}
					return reparse(fr);
				};
				bool ok = true;
				for (uint32_t i = 0; i < lo && ok; ++i) {
					Frag c1;
					ok = next_copy(&c1);
					if (ok) rep = cat(std::move(rep), std::move(c1));
					if (prog_.size() > GYS_RE_MAX_PROG) ok = bad("pattern too large");
				}
				if (ok && hi == ~0u) {
					Frag c1;
					ok = next_copy(&c1);
					if (ok) rep = cat(std::move(rep), star(std::move(c1)));
				} else if (ok && hi > lo) {
					// x{0,k} = (x(x(x)?)?)?  built from the inside out
					Frag tail;
					for (uint32_t i = 0; i < hi - lo && ok; ++i) {
						Frag c1;
						ok = next_copy(&c1);
						if (ok) tail = quest(cat(std::move(c1), std::move(tail)));
						if (prog_.size() > GYS_RE_MAX_PROG) ok = bad("pattern too large");
					}
					if (ok) rep = cat(std::move(rep), std::move(tail));
				} else if (ok && !first_used) {
					// x{0}: the operand's instructions stay unreachable
				}
				rep_mult_ = mult_outer;
				if (!ok) return false;
				p_ = resume;
				acc = cat(std::move(acc), std::move(rep));
			} else {
				acc = cat(std::move(acc), std::move(atom));
			}
			if (prog_.size() > GYS_RE_MAX_PROG) return bad("pattern too large");
		}
		*out = std::move(acc);
		return true;
	}
	bool parse_flags(Flags &f, bool *is_group) // after "(?": i s m U and their negations up to ')' or ':'
	{
		bool neg = false, any = false;
		while (p_ < end_) {
			const char c = *p_++;
			if (c == ')') {
				*is_group = false;
				return any || bad("missing flags");
			}
			if (c == ':') {
				*is_group = true;
				return true;
			}
			if (c == '-') {
				if (neg) return bad("bad flags");
				neg = true;
				continue;
			}
			any = true;
			if (c == 'i') f.icase = !neg;
			else if (c == 's') f.dotnl = !neg;
			else if (c == 'm') f.multiline = !neg;
			else if (c == 'U') { /* swaps greedy / lazy: same language */ }
			else return bad("unknown flag");
		}
		return bad("missing )");
	}
	static int hexval(char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }
	// an escape behind '\\': either a single byte (*byte >= 0) or a class added to cs (*byte = -1) or an assertion (*as >= 0; not inside [...])
	bool parse_escape(bool in_class, int *byte, CharSet *cs, int *as)
	{
		*byte = -1;
		*as = -1;
		if (p_ >= end_) return bad("trailing \\");
		const char c = *p_++;
		switch (c) {
		case 'd': cs->add_range('0', '9'); return true;
		case 'D': { CharSet t; t.add_range('0', '9'); t.negate(); cs->merge(t); return true; }
		case 'w': cs->add_range('0', '9'); cs->add_range('A', 'Z'); cs->add_range('a', 'z'); cs->add('_'); return true;
		case 'W': { CharSet t; t.add_range('0', '9'); t.add_range('A', 'Z'); t.add_range('a', 'z'); t.add('_'); t.negate(); cs->merge(t); return true; }
		case 's': cs->add(' '); cs->add('\t'); cs->add('\n'); cs->add('\f'); cs->add('\r'); return true; // (RE2's \s: no \v)
		case 'S': { CharSet t; t.add(' '); t.add('\t'); t.add('\n'); t.add('\f'); t.add('\r'); t.negate(); cs->merge(t); return true; }
		case 'n': *byte = '\n'; return true;
		case 't': *byte = '\t'; return true;
		case 'r': *byte = '\r'; return true;
		case 'f': *byte = '\f'; return true;
		case 'v': *byte = '\v'; return true;
		case 'a': *byte = '\a'; return true;
		case '0': *byte = 0; return true;
		case 'x': {
			if (p_ < end_ && *p_ == '{') {
				++p_;
				int v = 0, nd = 0;
				while (p_ < end_ && *p_ != '}') {
					const int h = hexval(*p_++);
					if (h < 0 || ++nd > 2) return bad("bad \\x{..} (bytes only)");
					v = v * 16 + h;
				}
				if (p_ >= end_ || !nd) return bad("bad \\x{..}");
				++p_;
				*byte = v;
				return true;
			}
			if (end_ - p_ < 2 || hexval(p_[0]) < 0 || hexval(p_[1]) < 0) return bad("bad \\x escape");
			*byte = hexval(p_[0]) * 16 + hexval(p_[1]);
			p_ += 2;
			return true;
		}
		case 'b': if (in_class) return bad("\\b inside a class"); *as = A_WORDB; return true;
		case 'B': if (in_class) return bad("\\B inside a class"); *as = A_NWORDB; return true;
		case 'A': if (in_class) return bad("\\A inside a class"); *as = A_BOT; return true;
		case 'z': if (in_class) return bad("\\z inside a class"); *as = A_EOT; return true;
		default:
			if ((c >= '1' && c <= '9')) return bad("back-references are not supported");
			if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) return bad("unsupported escape");
			*byte = (unsigned char)c; // escaped punctuation
			return true;
		}
	}
	bool parse_class(const Flags &f, Frag *out)
	{
		CharSet cs;
		bool neg = false;
		if (p_ < end_ && *p_ == '^') {
			neg = true;
			++p_;
		}
		bool first = true;
		for (;;) {
			if (p_ >= end_) return bad("missing ]");
			char c = *p_;
			if (c == ']' && !first) {
				++p_;
				break;
			}
			first = false;
			if (c == '[' && p_ + 1 < end_ && p_[1] == ':') {
				const char *q = p_ + 2;
				bool pneg = false;
				if (q < end_ && *q == '^') {
					pneg = true;
					++q;
				}
				const char *name = q;
				while (q < end_ && *q != ':') ++q;
				if (q + 1 >= end_ || q[1] != ']') return bad("bad [[:class:]]");
				const std::string nm(name, q);
				CharSet t;
				if (nm == "alpha") { t.add_range('A', 'Z'); t.add_range('a', 'z'); }
				else if (nm == "digit") t.add_range('0', '9');
				else if (nm == "alnum") { t.add_range('0', '9'); t.add_range('A', 'Z'); t.add_range('a', 'z'); }
				else if (nm == "upper") t.add_range('A', 'Z');
				else if (nm == "lower") t.add_range('a', 'z');
				else if (nm == "space") { t.add(' '); t.add_range('\t', '\r'); }
				else if (nm == "blank") { t.add(' '); t.add('\t'); }
				else if (nm == "punct") { t.add_range('!', '/'); t.add_range(':', '@'); t.add_range('[', '`'); t.add_range('{', '~'); }
				else if (nm == "xdigit") { t.add_range('0', '9'); t.add_range('A', 'F'); t.add_range('a', 'f'); }
				else if (nm == "word") { t.add_range('0', '9'); t.add_range('A', 'Z'); t.add_range('a', 'z'); t.add('_'); }
				else if (nm == "cntrl") { t.add_range(0, 31); t.add(127); }
				else if (nm == "print") t.add_range(' ', '~');
				else if (nm == "graph") t.add_range('!', '~');
				else if (nm == "ascii") t.add_range(0, 127);
				else return bad("unknown [[:class:]]");
				if (pneg) t.negate();
				cs.merge(t);
				p_ = q + 2;
				continue;
			}
			int lo;
			++p_;
			if (c == '\\') {
				int as;
				CharSet t;
				if (!parse_escape(true, &lo, &t, &as)) return false;
				if (lo < 0) {
					cs.merge(t);
					continue;
				}
			} else {
				lo = (unsigned char)c;
			}
			int hi = lo;
			if (p_ + 1 < end_ && *p_ == '-' && p_[1] != ']') {
				++p_;
				char d = *p_++;
				if (d == '\\') {
					int as;
					CharSet t;
					if (!parse_escape(true, &hi, &t, &as)) return false;
					if (hi < 0) return bad("bad character class range");
				} else {
					hi = (unsigned char)d;
				}
				if (hi < lo) return bad("bad character class range");
			}
			cs.add_range((unsigned)lo, (unsigned)hi);
		}
		if (f.icase) cs.fold_case();
		if (neg) cs.negate();
		Flags nf = f;
		nf.icase = false; // (already folded; a negated class must not be folded again)
		*out = frag_set(cs, nf);
		return true;
	}
	bool parse_atom(Flags &f, int depth, Frag *out, bool *flag_only)
	{
		// every copy of a counted repetition re-parses its operand: the work is bounded here, whatever the operands emit (an empty group
		// emits nothing, so the program-size limit alone does not stop ((((){1000}){1000}){1000}){1000})
		if (++atoms_parsed_ > GYS_RE_MAX_ATOMS) return bad("pattern too large");
		*flag_only = false;
		const char c = *p_++;
		switch (c) {
		case '(': {
			Flags gf = f;
			if (p_ < end_ && *p_ == '?') {
				++p_;
				if (p_ < end_ && *p_ == 'P') { // (?P<name>...)
					++p_;
					if (p_ >= end_ || *p_ != '<') return bad("bad named group");
					while (p_ < end_ && *p_ != '>') ++p_;
					if (p_ >= end_) return bad("bad named group");
					++p_;
				} else if (p_ < end_ && (*p_ == '=' || *p_ == '!' || *p_ == '<')) {
					return bad("look-around is not supported");
				} else {
					bool is_group = false;
					if (!parse_flags(gf, &is_group)) return false;
					if (!is_group) {
						f = gf; // (?i): applies to the rest of the enclosing group
						*flag_only = true;
						return true;
					}
				}
			}
			Frag inner;
			if (!parse_alt(gf, depth + 1, &inner)) return false;
			if (p_ >= end_ || *p_ != ')') return bad("missing )");
			++p_;
			*out = std::move(inner);
			return true;
		}
		case '[': return parse_class(f, out);
		case '.': {
			CharSet cs;
			cs.add_range(0, 255);
			if (!f.dotnl) cs.w[0] &= ~(1ull << '\n');
			Flags nf = f;
			nf.icase = false;
			*out = frag_set(cs, nf);
			return true;
		}
		case '^': *out = frag_assert(f.multiline ? A_BOL : A_BOT); return true;
		case '$': *out = frag_assert(f.multiline ? A_EOL : A_EOT); return true;
		case '*': case '+': case '?': return bad("missing argument to repetition operator");
		case '\\': {
			int byte, as;
			CharSet cs;
			if (!parse_escape(false, &byte, &cs, &as)) return false;
			if (as >= 0) {
				*out = frag_assert((Assert)as);
				return true;
			}
			if (byte >= 0) cs.add((unsigned)byte);
			Flags nf = f;
			if (byte < 0) nf.icase = false; // (\d, \W ...: folding a negated class would widen it)
			*out = frag_set(cs, nf);
			return true;
		}
		default: {
			CharSet cs;
			cs.add((unsigned char)c);
			*out = frag_set(cs, f);
			return true;
		}
		}
	}
};

} // namespace gysre
