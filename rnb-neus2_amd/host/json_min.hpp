// json_min.hpp — small JSON reader for the testbed host (transform.json, configs/nerf/base.json).
// Accepts // and /* */ comments like the reference's nlohmann::json::parse(..., ignore_comments = true)
// (src/nerf_loader.cpp:236, src/testbed.cu:63-74).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace jsonmin {

struct Value;
using Array = std::vector<Value>;
using Object = std::map<std::string, Value>;

struct Value {
	enum Type { Null, Bool, Number, String, Arr, Obj } type = Null;
	bool b = false;
	double num = 0.0;
	std::string str;
	std::shared_ptr<Array> arr;
	std::shared_ptr<Object> obj;

	bool is_null() const { return type == Null; }
	bool is_array() const { return type == Arr; }
	bool is_object() const { return type == Obj; }
	bool is_number() const { return type == Number; }
	bool is_string() const { return type == String; }
	bool is_bool() const { return type == Bool; }
	bool contains(const std::string& k) const { return type == Obj && obj->count(k) > 0; }
	size_t size() const { return type == Arr ? arr->size() : type == Obj ? obj->size() : 0; }
	const Value& operator[](const std::string& k) const {
		static const Value null_value;
		if (type != Obj) return null_value;
		auto it = obj->find(k);
		return it == obj->end() ? null_value : it->second;
	}
	const Value& operator[](size_t i) const {
		if (type != Arr || i >= arr->size()) throw std::runtime_error("json: array index out of range");
		return (*arr)[i];
	}
	double as_number() const {
		if (type == Number) return num;
		if (type == Bool) return b ? 1.0 : 0.0;
		throw std::runtime_error("json: value is not a number");
	}
	float as_float() const { return (float)as_number(); }
	int64_t as_int() const { return (int64_t)as_number(); }
	bool as_bool() const {
		if (type == Bool) return b;
		if (type == Number) return num != 0.0;
		throw std::runtime_error("json: value is not a bool");
	}
	const std::string& as_string() const {
		if (type != String) throw std::runtime_error("json: value is not a string");
		return str;
	}
	template <typename T> T value(const std::string& k, T def) const {
		if (!contains(k)) return def;
		const Value& v = (*this)[k];
		if (v.is_null()) return def;
		if constexpr (std::is_same<T, bool>::value) return v.as_bool();
		else if constexpr (std::is_same<T, std::string>::value) return v.as_string();
		else return (T)v.as_number();
	}
};

class Parser {
public:
	explicit Parser(const std::string& s) : s_(s) {}
	Value parse() {
		Value v = value();
		ws();
		if (p_ != s_.size()) fail("trailing characters");
		return v;
	}
private:
	const std::string& s_;
	size_t p_ = 0;
	[[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("json parse error at byte ") + std::to_string(p_) + ": " + what); }
	void ws() {
		for (;;) {
			while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\t' || s_[p_] == '\n' || s_[p_] == '\r')) ++p_;
			if (p_ + 1 < s_.size() && s_[p_] == '/' && s_[p_ + 1] == '/') { while (p_ < s_.size() && s_[p_] != '\n') ++p_; continue; }
			if (p_ + 1 < s_.size() && s_[p_] == '/' && s_[p_ + 1] == '*') { p_ += 2; while (p_ + 1 < s_.size() && !(s_[p_] == '*' && s_[p_ + 1] == '/')) ++p_; p_ += 2; continue; }
			break;
		}
	}
	Value value() {
		ws();
		if (p_ >= s_.size()) fail("unexpected end");
		char c = s_[p_];
		if (c == '{') return object();
		if (c == '[') return array();
		if (c == '"') { Value v; v.type = Value::String; v.str = string(); return v; }
		if (s_.compare(p_, 4, "true") == 0) { p_ += 4; Value v; v.type = Value::Bool; v.b = true; return v; }
		if (s_.compare(p_, 5, "false") == 0) { p_ += 5; Value v; v.type = Value::Bool; v.b = false; return v; }
		if (s_.compare(p_, 4, "null") == 0) { p_ += 4; return Value(); }
		if (s_.compare(p_, 3, "NaN") == 0) { p_ += 3; Value v; v.type = Value::Number; v.num = NAN; return v; }
		return number();
	}
	Value number() {
		const char* b = s_.c_str() + p_;
		char* e = nullptr;
		double d = std::strtod(b, &e);
		if (e == b) fail("invalid number");
		p_ += (size_t)(e - b);
		Value v; v.type = Value::Number; v.num = d; return v;
	}
	std::string string() {
		std::string out;
		++p_; // opening quote
		while (p_ < s_.size() && s_[p_] != '"') {
			char c = s_[p_++];
			if (c == '\\') {
				if (p_ >= s_.size()) fail("bad escape");
				char e = s_[p_++];
				switch (e) {
					case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
					case 'b': out += '\b'; break; case 'f': out += '\f'; break;
					case 'u': {
						if (p_ + 4 > s_.size()) fail("bad \\u escape");
						unsigned cp = (unsigned)std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
						p_ += 4;
						if (cp < 0x80) out += (char)cp;
						else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
						else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
						break;
					}
					default: out += e;
				}
			} else out += c;
		}
		if (p_ >= s_.size()) fail("unterminated string");
		++p_;
		return out;
	}
	Value array() {
		Value v; v.type = Value::Arr; v.arr = std::make_shared<Array>();
		++p_;
		ws();
		if (p_ < s_.size() && s_[p_] == ']') { ++p_; return v; }
		for (;;) {
			v.arr->push_back(value());
			ws();
			if (p_ >= s_.size()) fail("unterminated array");
			if (s_[p_] == ',') { ++p_; continue; }
			if (s_[p_] == ']') { ++p_; return v; }
			fail("expected , or ]");
		}
	}
	Value object() {
		Value v; v.type = Value::Obj; v.obj = std::make_shared<Object>();
		++p_;
		ws();
		if (p_ < s_.size() && s_[p_] == '}') { ++p_; return v; }
		for (;;) {
			ws();
			if (p_ >= s_.size() || s_[p_] != '"') fail("expected key");
			std::string k = string();
			ws();
			if (p_ >= s_.size() || s_[p_] != ':') fail("expected :");
			++p_;
			(*v.obj)[k] = value();
			ws();
			if (p_ >= s_.size()) fail("unterminated object");
			if (s_[p_] == ',') { ++p_; continue; }
			if (s_[p_] == '}') { ++p_; return v; }
			fail("expected , or }");
		}
	}
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }
inline Value parse_file(const std::string& path) {
	std::ifstream f(path);
	if (!f) throw std::runtime_error("cannot open " + path);
	std::stringstream ss;
	ss << f.rdbuf();
	std::string text = ss.str();
	return parse(text);
}

} // namespace jsonmin
