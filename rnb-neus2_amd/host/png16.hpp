// png16.hpp — PNG reader producing 4 x uint16 per pixel, with the conversions of stbi_load_16(..., desired_channels = 4)
// (what the reference's loader hands to the GPU, src/nerf_loader.cu:612,653): 8-bit samples are widened as v*257,
// grey is replicated to RGB, a missing alpha channel becomes 65535, palettes and tRNS keys are expanded.
// Non-interlaced and Adam7-interlaced images, bit depths 1-16. Needs zlib.
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace png16 {

struct Image { uint32_t width = 0, height = 0; std::vector<uint16_t> rgba; };

namespace detail {
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }

inline void unfilter(std::vector<uint8_t>& data, size_t offset, uint32_t rows, size_t stride, uint32_t bpp) {
	std::vector<uint8_t> zero(stride, 0);
	const uint8_t* prev = zero.data();
	for (uint32_t y = 0; y < rows; ++y) {
		uint8_t* line = data.data() + offset + (size_t)y * (stride + 1);
		const uint8_t ft = line[0];
		uint8_t* cur = line + 1;
		for (size_t i = 0; i < stride; ++i) {
			const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
			switch (ft) {
				case 0: break;
				case 1: cur[i] = (uint8_t)(cur[i] + a); break;
				case 2: cur[i] = (uint8_t)(cur[i] + b); break;
				case 3: cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1)); break;
				case 4: cur[i] = (uint8_t)(cur[i] + paeth(a, b, c)); break;
				default: throw std::runtime_error("png: bad filter type");
			}
		}
		prev = cur;
	}
}
} // namespace detail

inline Image load(const std::string& path) {
	using namespace detail;
	std::ifstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error("image not found: " + path);
	std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("not a PNG file: " + path);
	uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
	std::vector<uint8_t> idat, plte, trns;
	size_t p = 8;
	bool seen_ihdr = false;
	while (p + 12 <= file.size()) {
		const uint32_t len = be32(&file[p]);
		const std::string type((const char*)&file[p + 4], 4);
		if (p + 12 + (size_t)len > file.size()) throw std::runtime_error("png: truncated chunk in " + path);
		const uint8_t* d = &file[p + 8];
		if (type == "IHDR") {
			if (len < 13) throw std::runtime_error("png: bad IHDR");
			w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; seen_ihdr = true;
		} else if (type == "PLTE") plte.assign(d, d + len);
		else if (type == "tRNS") trns.assign(d, d + len);
		else if (type == "IDAT") idat.insert(idat.end(), d, d + len);
		else if (type == "IEND") break;
		p += 12 + (size_t)len;
	}
	if (!seen_ihdr || w == 0 || h == 0) throw std::runtime_error("png: missing IHDR in " + path);
	int channels;
	switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break; default: throw std::runtime_error("png: bad colour type"); }
	const uint32_t bits_pp = (uint32_t)channels * (uint32_t)depth;
	const uint32_t bpp = std::max(1u, bits_pp / 8);
	auto stride_of = [&](uint32_t pw) { return ((size_t)pw * bits_pp + 7) / 8; };
	// pass geometry (Adam7 or a single pass)
	struct Pass { uint32_t x0, y0, dx, dy; };
	std::vector<Pass> passes;
	if (interlace) passes = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
	else passes = {{0, 0, 1, 1}};
	size_t raw_size = 0;
	for (auto& ps : passes) {
		const uint32_t pw = (w > ps.x0) ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = (h > ps.y0) ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
		if (pw && ph) raw_size += (stride_of(pw) + 1) * ph;
	}
	std::vector<uint8_t> raw(raw_size);
	{
		uLongf dl = (uLongf)raw_size;
		int rc = uncompress(raw.data(), &dl, idat.data(), (uLong)idat.size());
		if (rc != Z_OK || dl != raw_size) throw std::runtime_error("png: inflate failed for " + path);
	}
	Image img; img.width = w; img.height = h; img.rgba.assign((size_t)w * h * 4, 0);
	auto widen = [&](uint32_t v) -> uint16_t { // sample of `depth` bits -> 16 bits; stb widens 8-bit as v*257 and scales <8-bit to 8-bit first
		switch (depth) {
			case 16: return (uint16_t)v;
			case 8: return (uint16_t)(v * 257u);
			case 4: return (uint16_t)((v * 17u) * 257u);
			case 2: return (uint16_t)((v * 85u) * 257u);
			default: return (uint16_t)((v * 255u) * 257u);
		}
	};
	size_t off = 0;
	for (auto& ps : passes) {
		const uint32_t pw = (w > ps.x0) ? (w - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = (h > ps.y0) ? (h - ps.y0 + ps.dy - 1) / ps.dy : 0;
		if (!pw || !ph) continue;
		const size_t stride = stride_of(pw);
		unfilter(raw, off, ph, stride, bpp);
		for (uint32_t py = 0; py < ph; ++py) {
			const uint8_t* line = raw.data() + off + (size_t)py * (stride + 1) + 1;
			for (uint32_t px = 0; px < pw; ++px) {
				uint32_t s[4] = {0, 0, 0, 0};
				for (int c = 0; c < channels; ++c) {
					const size_t idx = (size_t)px * channels + c;
					if (depth == 16) s[c] = ((uint32_t)line[idx * 2] << 8) | line[idx * 2 + 1];
					else if (depth == 8) s[c] = line[idx];
					else { const size_t bit = idx * depth; s[c] = (line[bit / 8] >> (8 - depth - (bit % 8))) & ((1u << depth) - 1u); }
				}
				uint16_t o[4];
				if (ctype == 3) {
					const uint32_t k = s[0];
					if ((size_t)k * 3 + 2 >= plte.size()) throw std::runtime_error("png: palette index out of range");
					o[0] = (uint16_t)(plte[k * 3] * 257u); o[1] = (uint16_t)(plte[k * 3 + 1] * 257u); o[2] = (uint16_t)(plte[k * 3 + 2] * 257u);
					o[3] = (uint16_t)((k < trns.size() ? trns[k] : 255u) * 257u);
				} else if (ctype == 0 || ctype == 4) {
					const uint16_t g = widen(s[0]);
					o[0] = o[1] = o[2] = g;
					if (ctype == 4) o[3] = widen(s[1]);
					else { o[3] = 65535; if (trns.size() >= 2 && s[0] == (((uint32_t)trns[0] << 8) | trns[1])) o[3] = 0; }
				} else {
					o[0] = widen(s[0]); o[1] = widen(s[1]); o[2] = widen(s[2]);
					if (ctype == 6) o[3] = widen(s[3]);
					else {
						o[3] = 65535;
						if (trns.size() >= 6 && s[0] == (((uint32_t)trns[0] << 8) | trns[1]) && s[1] == (((uint32_t)trns[2] << 8) | trns[3]) && s[2] == (((uint32_t)trns[4] << 8) | trns[5])) o[3] = 0;
					}
				}
				const uint32_t x = ps.x0 + px * ps.dx, y = ps.y0 + py * ps.dy;
				std::memcpy(&img.rgba[((size_t)y * w + x) * 4], o, 8);
			}
		}
		off += (stride + 1) * ph;
	}
	return img;
}

// IHDR only: dimensions plus the channel count / bit depth an "unchanged" decode would report
// (grey 1, grey+alpha 2, RGB / palette 3, RGBA / palette+tRNS 4; depth 8 for everything below 16 bits).
struct Info { uint32_t width = 0, height = 0; int channels = 0, depth = 0; };
inline Info probe(const std::string& path) {
	std::ifstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error("image not found: " + path);
	std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
	if (file.size() < 33 || std::memcmp(&file[1], "PNG", 3) != 0 || std::memcmp(&file[12], "IHDR", 4) != 0) throw std::runtime_error("not a PNG file: " + path);
	Info i; i.width = detail::be32(&file[16]); i.height = detail::be32(&file[20]);
	i.depth = file[24] == 16 ? 16 : 8;
	bool has_trns = false;
	for (size_t p = 8; p + 12 <= file.size();) {
		const uint32_t len = detail::be32(&file[p]);
		if (std::memcmp(&file[p + 4], "tRNS", 4) == 0) has_trns = true;
		if (std::memcmp(&file[p + 4], "IDAT", 4) == 0) break;
		p += 12 + (size_t)len;
	}
	switch (file[25]) { case 0: i.channels = has_trns ? 2 : 1; break; case 2: i.channels = has_trns ? 4 : 3; break; case 3: i.channels = has_trns ? 4 : 3; break; case 4: i.channels = 2; break; default: i.channels = 4; }
	return i;
}

// Writer: non-interlaced, filter 0 on every row, `channels` in {1,2,3,4} (grey, grey+alpha, RGB, RGBA), depth 8 or 16.
// `data` is row-major, interleaved, host-endian uint8 / uint16.
inline void save(const std::string& path, const void* data, uint32_t w, uint32_t h, int channels, int depth, int level = 1) {
	if ((depth != 8 && depth != 16) || channels < 1 || channels > 4) throw std::runtime_error("png: unsupported layout");
	static const int CT[5] = {0, 0, 4, 2, 6};
	const size_t row = (size_t)w * channels * (depth / 8);
	std::vector<uint8_t> raw((row + 1) * h);
	for (uint32_t y = 0; y < h; ++y) {
		uint8_t* o = &raw[(row + 1) * y];
		*o++ = 0;
		if (depth == 8) std::memcpy(o, (const uint8_t*)data + row * y, row);
		else { const uint16_t* s = (const uint16_t*)data + (size_t)w * channels * y; for (size_t k = 0; k < (size_t)w * channels; ++k) { o[2 * k] = (uint8_t)(s[k] >> 8); o[2 * k + 1] = (uint8_t)s[k]; } }
	}
	uLongf zl = compressBound((uLong)raw.size());
	std::vector<uint8_t> z(zl);
	if (compress2(z.data(), &zl, raw.data(), (uLong)raw.size(), level) != Z_OK) throw std::runtime_error("png: deflate failed");
	std::ofstream f(path, std::ios::binary);
	if (!f) throw std::runtime_error("cannot write " + path);
	auto chunk = [&](const char* tag, const uint8_t* d, size_t n) {
		uint8_t hd[8] = {(uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n, (uint8_t)tag[0], (uint8_t)tag[1], (uint8_t)tag[2], (uint8_t)tag[3]};
		uLong c = crc32(0L, hd + 4, 4);
		if (n) c = crc32(c, d, (uInt)n);
		const uint8_t cr[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
		f.write((const char*)hd, 8); if (n) f.write((const char*)d, (std::streamsize)n); f.write((const char*)cr, 4);
	};
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	f.write((const char*)sig, 8);
	const uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, (uint8_t)depth, (uint8_t)CT[channels], 0, 0, 0};
	chunk("IHDR", ihdr, 13);
	chunk("IDAT", z.data(), zl);
	chunk("IEND", nullptr, 0);
}

} // namespace png16
