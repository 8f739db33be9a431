// Small self-contained JSON + msgpack value model for the host side of libngp_hip.
// The reference uses nlohmann::json for configs, transforms.json and (via to_msgpack/from_msgpack) snapshots
// (src/testbed.cu:249-275, 5219-5283); this is the subset that path needs, with no third-party dependency.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mj {

struct Value {
	enum Type { Null, Bool, Int, UInt, Float, String, Binary, Array, Object };
	Type type = Null;
	bool b = false;
	int64_t i = 0;
	uint64_t u = 0;
	double f = 0.0;
	std::string s; // String or Binary payload
	std::vector<Value> arr;
	std::vector<std::pair<std::string, Value>> obj;

	Value() {}
	static Value make_bool(bool v) { Value r; r.type = Bool; r.b = v; return r; }
	static Value make_int(int64_t v) { Value r; r.type = Int; r.i = v; return r; }
	static Value make_uint(uint64_t v) { Value r; r.type = UInt; r.u = v; return r; }
	static Value make_float(double v) { Value r; r.type = Float; r.f = v; return r; }
	static Value make_string(const std::string& v) { Value r; r.type = String; r.s = v; return r; }
	static Value make_binary(const void* p, size_t n) { Value r; r.type = Binary; r.s.assign((const char*)p, n); return r; }
	static Value make_array() { Value r; r.type = Array; return r; }
	static Value make_object() { Value r; r.type = Object; return r; }

	bool is_null() const { return type == Null; }
	bool is_number() const { return type == Int || type == UInt || type == Float; }
	bool is_array() const { return type == Array; }
	bool is_object() const { return type == Object; }
	bool is_string() const { return type == String; }

	double num() const {
		switch (type) {
			case Int: return (double)i;
			case UInt: return (double)u;
			case Float: return f;
			case Bool: return b ? 1.0 : 0.0;
			default: throw std::runtime_error("json: value is not a number");
		}
	}
	int64_t integer() const {
		switch (type) {
			case Int: return i;
			case UInt: return (int64_t)u;
			case Float:
				if (!(f > -9.2e18 && f < 9.2e18)) throw std::runtime_error("json: number out of range for an integer"); // (NaN included; the cast would be undefined)
				return (int64_t)f;
			case Bool: return b ? 1 : 0;
			default: throw std::runtime_error("json: value is not an integer");
		}
	}
	bool boolean() const {
		if (type == Bool) return b;
		if (is_number()) return num() != 0.0;
		throw std::runtime_error("json: value is not a bool");
	}
	const std::string& str() const {
		if (type != String) throw std::runtime_error("json: value is not a string");
		return s;
	}
	const Value* find(const std::string& key) const {
		if (type != Object) return nullptr;
		for (auto& kv : obj) if (kv.first == key) return &kv.second;
		return nullptr;
	}
	bool contains(const std::string& key) const { return find(key) != nullptr; }
	const Value& at(const std::string& key) const {
		const Value* v = find(key);
		if (!v) throw std::runtime_error("json: missing key '" + key + "'");
		return *v;
	}
	const Value& at(size_t idx) const {
		if (type != Array || idx >= arr.size()) throw std::runtime_error("json: array index out of range");
		return arr[idx];
	}
	size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }
	Value& set(const std::string& key, Value v) {
		if (type != Object) { *this = make_object(); }
		for (auto& kv : obj) if (kv.first == key) { kv.second = std::move(v); return kv.second; }
		obj.emplace_back(key, std::move(v));
		return obj.back().second;
	}
	Value& operator[](const std::string& key) {
		if (type != Object) { *this = make_object(); }
		for (auto& kv : obj) if (kv.first == key) return kv.second;
		obj.emplace_back(key, Value());
		return obj.back().second;
	}
	void push(Value v) {
		if (type != Array) { *this = make_array(); }
		arr.push_back(std::move(v));
	}
	double value(const std::string& key, double dflt) const { const Value* v = find(key); return (v && v->is_number()) ? v->num() : dflt; }
	bool value(const std::string& key, bool dflt) const { const Value* v = find(key); return v ? v->boolean() : dflt; }
	std::string value(const std::string& key, const char* dflt) const { const Value* v = find(key); return (v && v->is_string()) ? v->s : std::string(dflt); }
};

// RFC 7386 merge patch: what the reference does for "parent" config inheritance (src/testbed.cu:86-97)
inline void merge_patch(Value& target, const Value& patch) {
	if (!patch.is_object()) { target = patch; return; }
	if (!target.is_object()) target = Value::make_object();
	for (auto& kv : patch.obj) {
		if (kv.second.is_null()) {
			for (size_t i = 0; i < target.obj.size(); ++i) if (target.obj[i].first == kv.first) { target.obj.erase(target.obj.begin() + i); break; }
		} else {
			merge_patch(target[kv.first], kv.second);
		}
	}
}

// ------------------------------------------------------------------------------------------- JSON text
class JsonParser {
public:
	JsonParser(const char* p, size_t n) : m_p(p), m_end(p + n) {}
	Value parse() {
		Value v = value();
		ws();
		if (m_p != m_end) fail("trailing characters");
		return v;
	}

private:
	const char* m_p;
	const char* m_end;
	int m_depth = 0; // arrays / objects open around the cursor: the parser recurses, and these files are untrusted input
	struct Nest {
		int& d;
		explicit Nest(int& depth) : d(depth) { if (++d > 256) throw std::runtime_error("json parse error: nesting too deep"); }
		~Nest() { --d; }
	};
	[[noreturn]] void fail(const char* msg) { throw std::runtime_error(std::string("json parse error: ") + msg); }
	void ws() {
		for (;;) {
			while (m_p < m_end && (*m_p == ' ' || *m_p == '\t' || *m_p == '\n' || *m_p == '\r')) ++m_p;
			// comments are accepted like nlohmann's ignore_comments=true (src/nerf_loader.cu:283)
			if (m_p + 1 < m_end && m_p[0] == '/' && m_p[1] == '/') { while (m_p < m_end && *m_p != '\n') ++m_p; continue; }
			if (m_p + 1 < m_end && m_p[0] == '/' && m_p[1] == '*') {
				m_p += 2;
				while (m_p + 1 < m_end && !(m_p[0] == '*' && m_p[1] == '/')) ++m_p;
				if (m_p + 1 >= m_end) fail("unterminated comment");
				m_p += 2;
				continue;
			}
			break;
		}
	}
	Value value() {
		ws();
		if (m_p >= m_end) fail("unexpected end");
		char c = *m_p;
		if (c == '{') return object();
		if (c == '[') return array();
		if (c == '"') return Value::make_string(string());
		if (c == 't') { expect("true"); return Value::make_bool(true); }
		if (c == 'f') { expect("false"); return Value::make_bool(false); }
		if (c == 'n') { expect("null"); return Value(); }
		return number();
	}
	void expect(const char* lit) {
		size_t n = strlen(lit);
		if ((size_t)(m_end - m_p) < n || strncmp(m_p, lit, n) != 0) fail("bad literal");
		m_p += n;
	}
	Value number() {
		const char* start = m_p;
		bool is_float = false;
		if (m_p < m_end && (*m_p == '-' || *m_p == '+')) ++m_p;
		while (m_p < m_end && ((*m_p >= '0' && *m_p <= '9') || *m_p == '.' || *m_p == 'e' || *m_p == 'E' || *m_p == '-' || *m_p == '+')) {
			if (*m_p == '.' || *m_p == 'e' || *m_p == 'E') is_float = true;
			++m_p;
		}
		if (m_p == start) fail("bad number");
		std::string tok(start, m_p);
		if (!is_float) {
			try {
				if (tok[0] == '-') return Value::make_int(std::stoll(tok));
				return Value::make_uint(std::stoull(tok));
			} catch (...) { /* fall through to double */ }
		}
		return Value::make_float(std::stod(tok));
	}
	static void utf8(std::string& out, uint32_t cp) {
		if (cp < 0x80) out.push_back((char)cp);
		else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
		else if (cp < 0x10000) { out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
		else { out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F))); }
	}
	std::string string() {
		++m_p; // opening quote
		std::string out;
		while (m_p < m_end && *m_p != '"') {
			char c = *m_p++;
			if (c != '\\') { out.push_back(c); continue; }
			if (m_p >= m_end) fail("bad escape");
			char e = *m_p++;
			switch (e) {
				case '"': out.push_back('"'); break;
				case '\\': out.push_back('\\'); break;
				case '/': out.push_back('/'); break;
				case 'b': out.push_back('\b'); break;
				case 'f': out.push_back('\f'); break;
				case 'n': out.push_back('\n'); break;
				case 'r': out.push_back('\r'); break;
				case 't': out.push_back('\t'); break;
				case 'u': {
					if (m_end - m_p < 4) fail("bad \\u escape");
					uint32_t cp = (uint32_t)std::stoul(std::string(m_p, m_p + 4), nullptr, 16);
					m_p += 4;
					utf8(out, cp);
				} break;
				default: fail("bad escape");
			}
		}
		if (m_p >= m_end) fail("unterminated string");
		++m_p;
		return out;
	}
	Value array() {
		Nest nest(m_depth);
		++m_p;
		Value v = Value::make_array();
		ws();
		if (m_p < m_end && *m_p == ']') { ++m_p; return v; }
		for (;;) {
			v.arr.push_back(value());
			ws();
			if (m_p >= m_end) fail("unterminated array");
			if (*m_p == ',') { ++m_p; continue; }
			if (*m_p == ']') { ++m_p; return v; }
			fail("expected , or ]");
		}
	}
	Value object() {
		Nest nest(m_depth);
		++m_p;
		Value v = Value::make_object();
		ws();
		if (m_p < m_end && *m_p == '}') { ++m_p; return v; }
		for (;;) {
			ws();
			if (m_p >= m_end || *m_p != '"') fail("expected key");
			std::string key = string();
			ws();
			if (m_p >= m_end || *m_p != ':') fail("expected :");
			++m_p;
			v.obj.emplace_back(std::move(key), value());
			ws();
			if (m_p >= m_end) fail("unterminated object");
			if (*m_p == ',') { ++m_p; continue; }
			if (*m_p == '}') { ++m_p; return v; }
			fail("expected , or }");
		}
	}
};

inline Value parse_json(const std::string& text) { return JsonParser(text.data(), text.size()).parse(); }

// ------------------------------------------------------------------------------------------- msgpack
class MsgpackReader {
public:
	MsgpackReader(const uint8_t* p, size_t n) : m_p(p), m_end(p + n) {}
	Value parse() { return value(0); }
	size_t remaining() const { return (size_t)(m_end - m_p); }

private:
	const uint8_t* m_p;
	const uint8_t* m_end;
	void need(size_t n) { if ((size_t)(m_end - m_p) < n) throw std::runtime_error("msgpack: truncated input"); }
	uint64_t be(int n) {
		need((size_t)n);
		uint64_t v = 0;
		for (int k = 0; k < n; ++k) v = (v << 8) | *m_p++;
		return v;
	}
	std::string bytes(size_t n) {
		need(n);
		std::string s((const char*)m_p, n);
		m_p += n;
		return s;
	}
	Value array(size_t n, int depth) {
		Value v = Value::make_array();
		v.arr.reserve(n < 65536 ? n : 65536);
		for (size_t k = 0; k < n; ++k) v.arr.push_back(value(depth + 1));
		return v;
	}
	Value map(size_t n, int depth) {
		Value v = Value::make_object();
		for (size_t k = 0; k < n; ++k) {
			Value key = value(depth + 1);
			if (key.type != Value::String) throw std::runtime_error("msgpack: non-string map key");
			v.obj.emplace_back(key.s, value(depth + 1));
		}
		return v;
	}
	Value value(int depth) {
		if (depth > 128) throw std::runtime_error("msgpack: nesting too deep");
		need(1);
		uint8_t t = *m_p++;
		if (t <= 0x7f) return Value::make_uint(t);
		if (t >= 0xe0) return Value::make_int((int8_t)t);
		if ((t & 0xf0) == 0x80) return map(t & 0x0f, depth);
		if ((t & 0xf0) == 0x90) return array(t & 0x0f, depth);
		if ((t & 0xe0) == 0xa0) return Value::make_string(bytes(t & 0x1f));
		switch (t) {
			case 0xc0: return Value();
			case 0xc2: return Value::make_bool(false);
			case 0xc3: return Value::make_bool(true);
			case 0xc4: { size_t n = be(1); std::string s = bytes(n); return Value::make_binary(s.data(), s.size()); }
			case 0xc5: { size_t n = be(2); std::string s = bytes(n); return Value::make_binary(s.data(), s.size()); }
			case 0xc6: { size_t n = be(4); need(n); Value v = Value::make_binary(m_p, n); m_p += n; return v; }
			case 0xc7: { size_t n = be(1); be(1); bytes(n); return Value(); }
			case 0xc8: { size_t n = be(2); be(1); bytes(n); return Value(); }
			case 0xc9: { size_t n = be(4); be(1); bytes(n); return Value(); }
			case 0xca: { uint32_t u = (uint32_t)be(4); float f; memcpy(&f, &u, 4); return Value::make_float(f); }
			case 0xcb: { uint64_t u = be(8); double d; memcpy(&d, &u, 8); return Value::make_float(d); }
			case 0xcc: return Value::make_uint(be(1));
			case 0xcd: return Value::make_uint(be(2));
			case 0xce: return Value::make_uint(be(4));
			case 0xcf: return Value::make_uint(be(8));
			case 0xd0: return Value::make_int((int8_t)be(1));
			case 0xd1: return Value::make_int((int16_t)be(2));
			case 0xd2: return Value::make_int((int32_t)be(4));
			case 0xd3: return Value::make_int((int64_t)be(8));
			case 0xd4: be(1); bytes(1); return Value();
			case 0xd5: be(1); bytes(2); return Value();
			case 0xd6: be(1); bytes(4); return Value();
			case 0xd7: be(1); bytes(8); return Value();
			case 0xd8: be(1); bytes(16); return Value();
			case 0xd9: return Value::make_string(bytes(be(1)));
			case 0xda: return Value::make_string(bytes(be(2)));
			case 0xdb: return Value::make_string(bytes(be(4)));
			case 0xdc: return array(be(2), depth);
			case 0xdd: return array(be(4), depth);
			case 0xde: return map(be(2), depth);
			case 0xdf: return map(be(4), depth);
			default: throw std::runtime_error("msgpack: unsupported type byte");
		}
	}
};

class MsgpackWriter {
public:
	std::string out;
	void write(const Value& v) {
		switch (v.type) {
			case Value::Null: put(0xc0); break;
			case Value::Bool: put(v.b ? 0xc3 : 0xc2); break;
			case Value::UInt: uint_(v.u); break;
			case Value::Int: if (v.i >= 0) uint_((uint64_t)v.i); else int_(v.i); break;
			case Value::Float: {
				// like nlohmann: float32 when the value survives the round trip, else float64
				float f = (float)v.f;
				if ((double)f == v.f) { uint32_t u; memcpy(&u, &f, 4); put(0xca); be(u, 4); }
				else { uint64_t u; memcpy(&u, &v.f, 8); put(0xcb); be(u, 8); }
			} break;
			case Value::String: {
				size_t n = v.s.size();
				if (n <= 31) put((uint8_t)(0xa0 | n));
				else if (n <= 0xff) { put(0xd9); be(n, 1); }
				else if (n <= 0xffff) { put(0xda); be(n, 2); }
				else { put(0xdb); be(n, 4); }
				out.append(v.s);
			} break;
			case Value::Binary: {
				size_t n = v.s.size();
				if (n <= 0xff) { put(0xc4); be(n, 1); }
				else if (n <= 0xffff) { put(0xc5); be(n, 2); }
				else { put(0xc6); be(n, 4); }
				out.append(v.s);
			} break;
			case Value::Array: {
				size_t n = v.arr.size();
				if (n <= 15) put((uint8_t)(0x90 | n));
				else if (n <= 0xffff) { put(0xdc); be(n, 2); }
				else { put(0xdd); be(n, 4); }
				for (auto& e : v.arr) write(e);
			} break;
			case Value::Object: {
				size_t n = v.obj.size();
				if (n <= 15) put((uint8_t)(0x80 | n));
				else if (n <= 0xffff) { put(0xde); be(n, 2); }
				else { put(0xdf); be(n, 4); }
				for (auto& kv : v.obj) { write(Value::make_string(kv.first)); write(kv.second); }
			} break;
		}
	}

private:
	void put(uint8_t b) { out.push_back((char)b); }
	void be(uint64_t v, int n) { for (int k = n - 1; k >= 0; --k) put((uint8_t)(v >> (8 * k))); }
	void uint_(uint64_t u) {
		if (u <= 0x7f) put((uint8_t)u);
		else if (u <= 0xff) { put(0xcc); be(u, 1); }
		else if (u <= 0xffff) { put(0xcd); be(u, 2); }
		else if (u <= 0xffffffffull) { put(0xce); be(u, 4); }
		else { put(0xcf); be(u, 8); }
	}
	void int_(int64_t i) {
		if (i >= -32) put((uint8_t)(int8_t)i);
		else if (i >= -128) { put(0xd0); be((uint64_t)(uint8_t)(int8_t)i, 1); }
		else if (i >= -32768) { put(0xd1); be((uint64_t)(uint16_t)(int16_t)i, 2); }
		else if (i >= -2147483648ll) { put(0xd2); be((uint64_t)(uint32_t)(int32_t)i, 4); }
		else { put(0xd3); be((uint64_t)i, 8); }
	}
};

} // namespace mj
