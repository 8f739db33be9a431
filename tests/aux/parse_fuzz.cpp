// Fuzz harness for the snapshot / config / transforms parsers (csrc/minijson.h: JSON text, msgpack, the msgpack writer), built by
// tests/test_host_cpu.py with -fsanitize=address,undefined: reads a seed file (.json or .msgpack), applies `n` deterministic mutations and
// parses each variant, walks the result the way the loaders do (find / at / value / num / integer on every node) and, for what parsed,
// writes it back and parses it again. Any outcome but a crash, a hang or a sanitizer report is fine: these files are untrusted input.
#include "../../surface-irradiance-estimation-from-neural-radiance-fields_amd/csrc/minijson.h"

#include <cstdio>
#include <fstream>
#include <sstream>



static uint64_t walk(const mj::Value& v, int depth) {
	uint64_t h = (uint64_t)v.type;
	if (depth > 200) return h;
	if (v.is_number()) {
		h += (uint64_t)(v.num() != 0.0);
		try { h += (uint64_t)v.integer(); } catch (const std::exception&) {}
	}
	if (v.is_string()) h += v.s.size();
	if (v.is_array()) {
		for (size_t i = 0; i < v.size(); ++i) h = h * 31u + walk(v.at(i), depth + 1);
	}
	if (v.is_object()) {
		for (auto& kv : v.obj) h = h * 131u + kv.first.size() + walk(kv.second, depth + 1);
		h += (uint64_t)v.value("n_levels", 16.0) + (uint64_t)v.value("otype", "x").size() + (uint64_t)v.contains("snapshot");
	}
	return h;
}

int main(int argc, char** argv) {
	if (argc < 3) return 2;
	std::ifstream f(argv[1], std::ios::binary);
	std::stringstream ss;
	ss << f.rdbuf();
	const std::string seed = ss.str();
	const bool json = seed.size() && (seed[0] == '{' || seed[0] == '[' || seed[0] == ' ' || seed[0] == '\n');
	const int n = atoi(argv[2]);
	uint64_t state = 0x9E3779B97F4A7C15ull, sink = 0;
	auto rnd = [&]() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return state; };
	int ok = 0, refused = 0;
	for (int it = 0; it < n; ++it) {
		std::string v = seed;
		const int kind = (int)(rnd() % 5);
		if (kind == 0) v.resize((size_t)(rnd() % (v.size() + 1)));
		else if (kind == 1) { for (int k = 0; k < 1 + (int)(rnd() % 8); ++k) v[(size_t)(rnd() % v.size())] = (char)rnd(); }
		else if (kind == 2) { const size_t p = (size_t)(rnd() % v.size()); for (size_t q = p; q < p + 4 && q < v.size(); ++q) v[q] = (char)0xFF; }
		else if (kind == 3) { const size_t p = (size_t)(rnd() % v.size()); v.insert(p, std::string((size_t)(rnd() % 64), (char)rnd())); }
		else { const size_t p = (size_t)(rnd() % v.size()); v.insert(p, std::string((size_t)(rnd() % 3000), json ? '[' : (char)0x91)); } // deep nesting
		try {
			mj::Value r = json ? mj::parse_json(v) : mj::MsgpackReader((const uint8_t*)v.data(), v.size()).parse();
			sink += walk(r, 0);
			mj::MsgpackWriter wr;
			wr.write(r);
			mj::Value again = mj::MsgpackReader((const uint8_t*)wr.out.data(), wr.out.size()).parse();
			if (walk(again, 0) != walk(r, 0)) { printf("round trip mismatch\n"); return 1; }
			++ok;
		} catch (const std::exception&) {
			++refused;
		}
	}
	printf("parsed %d refused %d (%llu)\n", ok, refused, (unsigned long long)(sink & 0xff));
	return 0;
}
