// PNG decoder for NerfDataset images (the reference uses stb_image, src/nerf_loader.cu:520-640): 8 / 16-bit grey,
// grey+alpha, RGB, RGBA and palette images, non-interlaced; zlib does the inflate. Output: RGBA8.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace ngp {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline bool decode_png(const std::string& bytes, std::vector<uint8_t>& rgba, int& width, int& height, std::string& why) {
	static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
	const uint8_t* d = (const uint8_t*)bytes.data();
	const size_t n = bytes.size();
	if (n < 33 || memcmp(d, sig, 8) != 0) { why = "not a PNG file"; return false; }
	uint32_t w = 0, h = 0;
	int depth = 0, color = 0, interlace = 0;
	std::vector<uint8_t> idat, palette, trns;
	size_t pos = 8;
	bool got_end = false;
	while (pos + 12 <= n && !got_end) {
		const uint32_t len = be32(d + pos);
		const uint8_t* type = d + pos + 4;
		const uint8_t* body = d + pos + 8;
		if (pos + 12 + (size_t)len > n) { why = "truncated PNG chunk"; return false; }
		if (!memcmp(type, "IHDR", 4) && len >= 13) {
			w = be32(body); h = be32(body + 4); depth = body[8]; color = body[9]; interlace = body[12];
		} else if (!memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
		else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
		else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
		else if (!memcmp(type, "IEND", 4)) got_end = true;
		pos += 12 + (size_t)len;
	}
	if (!w || !h || w > 32768 || h > 32768) { why = "bad PNG header"; return false; }
	if (interlace) { why = "interlaced PNG files are not supported"; return false; }
	if (depth != 8 && depth != 16) { why = "PNG bit depth " + std::to_string(depth) + " is not supported"; return false; }
	int channels;
	switch (color) {
		case 0: channels = 1; break;
		case 2: channels = 3; break;
		case 3: channels = 1; if (depth != 8) { why = "bad palette PNG"; return false; } break;
		case 4: channels = 2; break;
		case 6: channels = 4; break;
		default: why = "bad PNG colour type"; return false;
	}
	const size_t bpp = (size_t)channels * (depth / 8), stride = bpp * w;
	std::vector<uint8_t> raw((stride + 1) * h);
	uLongf raw_len = (uLongf)raw.size();
	if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) { why = "corrupt PNG data"; return false; }
	// unfilter in place (PNG specification, section 9)
	std::vector<uint8_t> prev(stride, 0);
	for (uint32_t y = 0; y < h; ++y) {
		uint8_t* line = raw.data() + (stride + 1) * y;
		const int filter = line[0];
		uint8_t* cur = line + 1;
		for (size_t x = 0; x < stride; ++x) {
			const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
			int pred = 0;
			switch (filter) {
				case 0: pred = 0; break;
				case 1: pred = a; break;
				case 2: pred = b; break;
				case 3: pred = (a + b) >> 1; break;
				case 4: { const int pq = a + b - c, pa = std::abs(pq - a), pb = std::abs(pq - b), pc = std::abs(pq - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
				default: why = "bad PNG filter"; return false;
			}
			cur[x] = (uint8_t)(cur[x] + pred);
		}
		memcpy(prev.data(), cur, stride);
	}
	rgba.resize((size_t)w * h * 4);
	const size_t step = depth / 8; // 16-bit samples: the high byte
	for (uint32_t y = 0; y < h; ++y) {
		const uint8_t* cur = raw.data() + (stride + 1) * y + 1;
		for (uint32_t x = 0; x < w; ++x) {
			const uint8_t* px = cur + (size_t)x * bpp;
			uint8_t* o = rgba.data() + ((size_t)y * w + x) * 4;
			switch (color) {
				case 0: o[0] = o[1] = o[2] = px[0]; o[3] = 255; break;
				case 2: o[0] = px[0]; o[1] = px[step]; o[2] = px[2 * step]; o[3] = 255; break;
				case 3: {
					const size_t i = px[0];
					if (3 * i + 2 >= palette.size()) { why = "PNG palette index out of range"; return false; }
					o[0] = palette[3 * i]; o[1] = palette[3 * i + 1]; o[2] = palette[3 * i + 2];
					o[3] = i < trns.size() ? trns[i] : 255;
					break;
				}
				case 4: o[0] = o[1] = o[2] = px[0]; o[3] = px[step]; break;
				default: o[0] = px[0]; o[1] = px[step]; o[2] = px[2 * step]; o[3] = px[3 * step]; break;
			}
		}
	}
	width = (int)w;
	height = (int)h;
	return true;
}

} // namespace ngp
