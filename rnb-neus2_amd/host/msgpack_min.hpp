// msgpack_min.hpp — the subset of MessagePack the snapshot files use (nlohmann::json::to_msgpack layout,
// src/testbed.cu:3280-3314): maps with string keys, arrays, str, bin (raw little-endian device bytes,
// tiny-cuda-nn/gpu_memory_json.h:36-58), integers, float32/float64, bool, nil.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mpk {

struct Value;
using Map = std::vector<std::pair<std::string, Value>>;
struct Value {
	enum Type { Nil, Bool, Int, UInt, Float, Str, Bin, Arr, MapT } type = Nil;
	bool b = false;
	int64_t i = 0;
	uint64_t u = 0;
	double f = 0;
	std::string s;
	std::vector<uint8_t> bin;
	std::vector<Value> arr;
	Map map;
	static Value boolean(bool v) { Value x; x.type = Bool; x.b = v; return x; }
	static Value uint(uint64_t v) { Value x; x.type = UInt; x.u = v; return x; }
	static Value real(double v) { Value x; x.type = Float; x.f = v; return x; }
	static Value str(const std::string& v) { Value x; x.type = Str; x.s = v; return x; }
	static Value binary(const void* p, size_t n) { Value x; x.type = Bin; x.bin.assign((const uint8_t*)p, (const uint8_t*)p + n); return x; }
	static Value object() { Value x; x.type = MapT; return x; }
	static Value array() { Value x; x.type = Arr; return x; }
	Value& set(const std::string& k, Value v) {
		for (auto& kv : map) if (kv.first == k) { kv.second = std::move(v); return kv.second; }
		map.emplace_back(k, std::move(v));
		return map.back().second;
	}
	const Value* find(const std::string& k) const {
		for (auto& kv : map) if (kv.first == k) return &kv.second;
		return nullptr;
	}
	const Value& at(const std::string& k) const {
		const Value* v = find(k);
		if (!v) throw std::runtime_error("snapshot: missing key '" + k + "'");
		return *v;
	}
	double number() const {
		switch (type) { case Int: return (double)i; case UInt: return (double)u; case Float: return f; case Bool: return b ? 1 : 0; default: throw std::runtime_error("snapshot: value is not a number"); }
	}
};

class Writer {
public:
	std::vector<uint8_t> out;
	void write(const Value& v) {
		switch (v.type) {
			case Value::Nil: put(0xc0); break;
			case Value::Bool: put(v.b ? 0xc3 : 0xc2); break;
			case Value::UInt: uint(v.u); break;
			case Value::Int: if (v.i >= 0) uint((uint64_t)v.i); else sint(v.i); break;
			case Value::Float: {
				const float f32 = (float)v.f;
				if ((double)f32 == v.f) { put(0xca); uint32_t b; std::memcpy(&b, &f32, 4); be(b, 4); }
				else { put(0xcb); uint64_t b; std::memcpy(&b, &v.f, 8); be(b, 8); }
				break;
			}
			case Value::Str: str(v.s); break;
			case Value::Bin: {
				const size_t n = v.bin.size();
				if (n < 256) { put(0xc4); put((uint8_t)n); } else if (n < 65536) { put(0xc5); be(n, 2); } else { put(0xc6); be(n, 4); }
				out.insert(out.end(), v.bin.begin(), v.bin.end());
				break;
			}
			case Value::Arr: {
				const size_t n = v.arr.size();
				if (n < 16) put((uint8_t)(0x90 | n)); else if (n < 65536) { put(0xdc); be(n, 2); } else { put(0xdd); be(n, 4); }
				for (auto& e : v.arr) write(e);
				break;
			}
			case Value::MapT: {
				const size_t n = v.map.size();
				if (n < 16) put((uint8_t)(0x80 | n)); else if (n < 65536) { put(0xde); be(n, 2); } else { put(0xdf); be(n, 4); }
				for (auto& kv : v.map) { str(kv.first); write(kv.second); }
				break;
			}
		}
	}
private:
	void put(uint8_t b) { out.push_back(b); }
	void be(uint64_t v, int n) { for (int k = n - 1; k >= 0; --k) put((uint8_t)(v >> (8 * k))); }
	void uint(uint64_t u) {
		if (u < 128) put((uint8_t)u); else if (u < 256) { put(0xcc); put((uint8_t)u); } else if (u < 65536) { put(0xcd); be(u, 2); }
		else if (u < (1ull << 32)) { put(0xce); be(u, 4); } else { put(0xcf); be(u, 8); }
	}
	void sint(int64_t i) {
		if (i >= -32) put((uint8_t)i); else if (i >= -128) { put(0xd0); put((uint8_t)i); } else if (i >= -32768) { put(0xd1); be((uint16_t)i, 2); }
		else if (i >= -2147483648ll) { put(0xd2); be((uint32_t)i, 4); } else { put(0xd3); be((uint64_t)i, 8); }
	}
	void str(const std::string& s) {
		const size_t n = s.size();
		if (n < 32) put((uint8_t)(0xa0 | n)); else if (n < 256) { put(0xd9); put((uint8_t)n); } else if (n < 65536) { put(0xda); be(n, 2); } else { put(0xdb); be(n, 4); }
		out.insert(out.end(), s.begin(), s.end());
	}
};

class Reader {
public:
	Reader(const uint8_t* p, size_t n) : p_(p), n_(n) {}
	Value read() {
		const uint8_t t = get();
		Value v;
		if (t < 0x80) { v.type = Value::UInt; v.u = t; return v; }
		if (t >= 0xe0) { v.type = Value::Int; v.i = (int8_t)t; return v; }
		if ((t & 0xf0) == 0x80) return map(t & 0x0f);
		if ((t & 0xf0) == 0x90) return array(t & 0x0f);
		if ((t & 0xe0) == 0xa0) return str(t & 0x1f);
		switch (t) {
			case 0xc0: return v;
			case 0xc2: return Value::boolean(false);
			case 0xc3: return Value::boolean(true);
			case 0xc4: return bin(be(1)); case 0xc5: return bin(be(2)); case 0xc6: return bin(be(4));
			case 0xca: { uint32_t b = (uint32_t)be(4); float f; std::memcpy(&f, &b, 4); return Value::real(f); }
			case 0xcb: { uint64_t b = be(8); double d; std::memcpy(&d, &b, 8); return Value::real(d); }
			case 0xcc: return Value::uint(be(1)); case 0xcd: return Value::uint(be(2)); case 0xce: return Value::uint(be(4)); case 0xcf: return Value::uint(be(8));
			case 0xd0: v.type = Value::Int; v.i = (int8_t)be(1); return v;
			case 0xd1: v.type = Value::Int; v.i = (int16_t)be(2); return v;
			case 0xd2: v.type = Value::Int; v.i = (int32_t)be(4); return v;
			case 0xd3: v.type = Value::Int; v.i = (int64_t)be(8); return v;
			case 0xd9: return str(be(1)); case 0xda: return str(be(2)); case 0xdb: return str(be(4));
			case 0xdc: return array(be(2)); case 0xdd: return array(be(4));
			case 0xde: return map(be(2)); case 0xdf: return map(be(4));
			case 0xc7: case 0xc8: case 0xc9: { // ext 8/16/32: skip payload, keep as binary
				const uint64_t n = be(t == 0xc7 ? 1 : t == 0xc8 ? 2 : 4); get(); return bin(n);
			}
			default: throw std::runtime_error("msgpack: unsupported type byte");
		}
	}
private:
	const uint8_t* p_; size_t n_; size_t o_ = 0;
	uint8_t get() { if (o_ >= n_) throw std::runtime_error("msgpack: truncated"); return p_[o_++]; }
	uint64_t be(int k) { uint64_t v = 0; for (int q = 0; q < k; ++q) v = (v << 8) | get(); return v; }
	Value str(uint64_t n) { if (o_ + n > n_) throw std::runtime_error("msgpack: truncated"); Value v = Value::str(std::string((const char*)p_ + o_, (size_t)n)); o_ += n; return v; }
	Value bin(uint64_t n) { if (o_ + n > n_) throw std::runtime_error("msgpack: truncated"); Value v = Value::binary(p_ + o_, (size_t)n); o_ += n; return v; }
	Value array(uint64_t n) { Value v = Value::array(); v.arr.reserve((size_t)n); for (uint64_t q = 0; q < n; ++q) v.arr.push_back(read()); return v; }
	Value map(uint64_t n) {
		Value v = Value::object();
		for (uint64_t q = 0; q < n; ++q) { Value k = read(); if (k.type != Value::Str) throw std::runtime_error("msgpack: non-string key"); Value val = read(); v.map.emplace_back(k.s, std::move(val)); }
		return v;
	}
};

inline void save(const std::string& path, const Value& v) {
	Writer w; w.write(v);
	std::ofstream f(path, std::ios::out | std::ios::binary);
	if (!f) throw std::runtime_error("cannot write " + path);
	f.write((const char*)w.out.data(), (std::streamsize)w.out.size());
}
inline Value load(const std::string& path) {
	std::ifstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error("cannot open " + path);
	std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	return Reader(d.data(), d.size()).read();
}

} // namespace mpk
