// Baseline / extended-sequential JPEG decoder for NerfDataset images (the reference decodes with stb_image inside
// load_nerf, src/nerf_loader.cu:520-640). Written from the JPEG standard (ITU T.81) and JFIF: Huffman-coded 8-bit
// samples, 1 (grey) or 3 (YCbCr) components, any sampling factors, restart intervals; float IDCT; the chroma planes of
// 2x subsampled images are interpolated with the 3:1 triangle filter (libjpeg's "fancy upsampling", which stb_image
// also applies). Progressive, arithmetic-coded, 12-bit and CMYK files are refused with a message. Output: RGBA8.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace ngp {
namespace jpeg_detail {

struct Huffman {
	uint8_t bits[17] = {};
	uint8_t symbols[256] = {};
	int mincode[17], maxcode[18], valptr[17];
	int16_t fast[512]; // 9-bit lookahead: (length << 8) | symbol, or -1
	bool present = false;
	void build() {
		int code = 0, k = 0;
		for (int i = 0; i < 512; ++i) fast[i] = -1;
		for (int len = 1; len <= 16; ++len) {
			valptr[len] = k;
			mincode[len] = code;
			for (int i = 0; i < bits[len]; ++i, ++k, ++code) {
				if (len <= 9) {
					const int first = code << (9 - len);
					for (int f = 0; f < (1 << (9 - len)); ++f) fast[first + f] = (int16_t)((len << 8) | symbols[k]);
				}
			}
			maxcode[len] = bits[len] ? code - 1 : -1;
			code <<= 1;
		}
		maxcode[17] = 0x7FFFFFFF;
		present = true;
	}
};

struct BitReader {
	const uint8_t* p;
	const uint8_t* end;
	uint32_t acc = 0;
	int n = 0;
	bool hit_marker = false;
	int padded = 0; // zero bytes fed past the end of the entropy-coded data
	void fill() {
		while (n <= 24) {
			int byte = 0;
			if (!hit_marker && p < end) {
				byte = *p;
				if (byte == 0xFF) {
					const int next = p + 1 < end ? p[1] : 0xD9;
					if (next == 0x00) p += 2;      // stuffed zero
					else { hit_marker = true; byte = 0; } // a marker ends the segment: feed zeros
				} else {
					++p;
				}
			}
			if (hit_marker || p >= end) ++padded;
			acc |= (uint32_t)byte << (24 - n);
			n += 8;
		}
	}
	int peek(int k) { fill(); return (int)(acc >> (32 - k)); }
	void skip(int k) { acc <<= k; n -= k; }
	int get(int k) { if (k == 0) return 0; const int v = peek(k); skip(k); return v; }
	void reset() { acc = 0; n = 0; hit_marker = false; padded = 0; }
};

inline int decode_symbol(BitReader& br, const Huffman& h) {
	const int look = br.peek(9);
	const int f = h.fast[look];
	if (f >= 0) { br.skip(f >> 8); return f & 255; }
	int code = br.peek(16);
	for (int len = 10; len <= 16; ++len) {
		const int c = code >> (16 - len);
		if (h.maxcode[len] >= 0 && c <= h.maxcode[len] && c >= h.mincode[len]) {
			br.skip(len);
			return h.symbols[h.valptr[len] + c - h.mincode[len]];
		}
	}
	return -1;
}
inline int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

static const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

inline void idct8x8(const float* in, uint8_t* out, int stride) {
	static float c[8][8];
	static bool init = false;
	if (!init) {
		for (int x = 0; x < 8; ++x)
			for (int u = 0; u < 8; ++u) c[x][u] = (u == 0 ? std::sqrt(0.125f) : 0.5f) * std::cos((2 * x + 1) * u * 3.14159265358979323846f / 16.0f);
		init = true;
	}
	float tmp[64];
	for (int v = 0; v < 8; ++v)
		for (int x = 0; x < 8; ++x) {
			float s = 0.f;
			for (int u = 0; u < 8; ++u) s += c[x][u] * in[v * 8 + u];
			tmp[v * 8 + x] = s;
		}
	for (int x = 0; x < 8; ++x)
		for (int y = 0; y < 8; ++y) {
			float s = 0.f;
			for (int v = 0; v < 8; ++v) s += c[y][v] * tmp[v * 8 + x];
			const int q = (int)std::lrintf(s + 128.0f);
			out[y * stride + x] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
		}
}

struct Component {
	int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
	int w = 0, hgt = 0; // plane size in samples (padded to whole MCUs)
	int pred = 0;
	std::vector<uint8_t> plane;
};

// upsample a plane by (fx, fy) in {1, 2}: 3:1 triangle filter per doubled axis, replication otherwise
inline std::vector<uint8_t> upsample(const std::vector<uint8_t>& src, int w, int h, int fx, int fy) {
	std::vector<uint8_t> cur = src;
	int cw = w, ch = h;
	if (fy == 2) {
		std::vector<uint16_t> rows((size_t)cw * ch * 2); // 3*near + far, scaled by 4
		for (int y = 0; y < ch * 2; ++y) {
			const int near = y / 2, far = (y & 1) ? std::min(near + 1, ch - 1) : std::max(near - 1, 0);
			for (int x = 0; x < cw; ++x) rows[(size_t)y * cw + x] = (uint16_t)(3 * cur[(size_t)near * cw + x] + cur[(size_t)far * cw + x]);
		}
		ch *= 2;
		if (fx == 2) {
			std::vector<uint8_t> out((size_t)cw * 2 * ch);
			for (int y = 0; y < ch; ++y) {
				const uint16_t* r = rows.data() + (size_t)y * cw;
				uint8_t* o = out.data() + (size_t)y * cw * 2;
				for (int x = 0; x < cw; ++x) {
					const int l = r[std::max(x - 1, 0)], m = r[x], rr = r[std::min(x + 1, cw - 1)];
					o[2 * x] = (uint8_t)((3 * m + l + 8) >> 4);
					o[2 * x + 1] = (uint8_t)((3 * m + rr + 7) >> 4);
				}
			}
			return out;
		}
		std::vector<uint8_t> out((size_t)cw * ch);
		for (size_t i = 0; i < out.size(); ++i) out[i] = (uint8_t)((rows[i] + 2) >> 2);
		return out;
	}
	if (fx == 2) {
		std::vector<uint8_t> out((size_t)cw * 2 * ch);
		for (int y = 0; y < ch; ++y) {
			const uint8_t* r = cur.data() + (size_t)y * cw;
			uint8_t* o = out.data() + (size_t)y * cw * 2;
			for (int x = 0; x < cw; ++x) {
				const int l = r[std::max(x - 1, 0)], m = r[x], rr = r[std::min(x + 1, cw - 1)];
				o[2 * x] = (uint8_t)((3 * m + l + 1) >> 2);
				o[2 * x + 1] = (uint8_t)((3 * m + rr + 2) >> 2);
			}
		}
		return out;
	}
	if (fx == 1 && fy == 1) return cur;
	std::vector<uint8_t> out((size_t)cw * fx * ch * fy);
	for (int y = 0; y < ch * fy; ++y)
		for (int x = 0; x < cw * fx; ++x) out[(size_t)y * cw * fx + x] = cur[(size_t)(y / fy) * cw + x / fx];
	return out;
}

} // namespace jpeg_detail

inline bool decode_jpeg(const std::string& bytes, std::vector<uint8_t>& rgba, int& width, int& height, std::string& why) {
	using namespace jpeg_detail;
	const uint8_t* d = (const uint8_t*)bytes.data();
	const size_t n = bytes.size();
	if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) { why = "not a JPEG file"; return false; }
	uint16_t qt[4][64] = {};
	bool have_qt[4] = {};
	Huffman hdc[4], hac[4];
	std::vector<Component> comps;
	int W = 0, H = 0, restart_interval = 0, hmax = 1, vmax = 1;
	bool have_frame = false, done = false;
	size_t pos = 2;
	auto be16 = [&](size_t p) { return (int)((d[p] << 8) | d[p + 1]); };
	while (pos + 4 <= n && !done) {
		if (d[pos] != 0xFF) { ++pos; continue; }
		const int marker = d[pos + 1];
		if (marker == 0xFF) { ++pos; continue; }
		pos += 2;
		if (marker == 0xD9) break;
		if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;
		if (pos + 2 > n) break;
		const int len = be16(pos);
		if (len < 2 || pos + (size_t)len > n) { why = "truncated JPEG segment"; return false; }
		const size_t seg = pos + 2, seg_end = pos + (size_t)len;
		switch (marker) {
			case 0xDB: { // DQT
				size_t p = seg;
				while (p < seg_end) {
					const int pq = d[p] >> 4, tq = d[p] & 15;
					++p;
					if (tq > 3 || p + (pq ? 128u : 64u) > seg_end) { why = "bad JPEG quantisation table"; return false; }
					for (int i = 0; i < 64; ++i) {
						qt[tq][i] = pq ? (uint16_t)be16(p + 2 * (size_t)i) : d[p + (size_t)i];
					}
					p += pq ? 128 : 64;
					have_qt[tq] = true;
				}
				break;
			}
			case 0xC4: { // DHT
				size_t p = seg;
				while (p + 17 <= seg_end) {
					const int tc = d[p] >> 4, th = d[p] & 15;
					if (tc > 1 || th > 3) { why = "bad JPEG Huffman table"; return false; }
					Huffman& h = tc ? hac[th] : hdc[th];
					int total = 0;
					h.bits[0] = 0;
					for (int i = 1; i <= 16; ++i) { h.bits[i] = d[p + (size_t)i]; total += h.bits[i]; }
					p += 17;
					if (total > 256 || p + (size_t)total > seg_end) { why = "bad JPEG Huffman table"; return false; }
					memcpy(h.symbols, d + p, (size_t)total);
					p += (size_t)total;
					h.build();
				}
				break;
			}
			case 0xC0: case 0xC1: { // SOF0 / SOF1
				if (d[seg] != 8) { why = "only 8-bit JPEG files are supported"; return false; }
				H = be16(seg + 1);
				W = be16(seg + 3);
				const int nc = d[seg + 5];
				if ((int64_t)W * H > (1ll << 28)) { why = "JPEG image too large"; return false; }
				if (W <= 0 || H <= 0 || (nc != 1 && nc != 3)) { why = nc == 4 ? "CMYK JPEG files are not supported" : "bad JPEG frame header"; return false; }
				comps.resize((size_t)nc);
				for (int i = 0; i < nc; ++i) {
					Component& c = comps[(size_t)i];
					c.id = d[seg + 6 + 3 * (size_t)i];
					c.h = d[seg + 7 + 3 * (size_t)i] >> 4;
					c.v = d[seg + 7 + 3 * (size_t)i] & 15;
					c.tq = d[seg + 8 + 3 * (size_t)i];
					if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) { why = "bad JPEG component"; return false; }
					hmax = std::max(hmax, c.h);
					vmax = std::max(vmax, c.v);
				}
				have_frame = true;
				break;
			}
			case 0xC2: why = "progressive JPEG files are not supported (re-save as baseline)"; return false;
			case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
				why = "lossless / hierarchical / arithmetic-coded JPEG files are not supported"; return false;
			case 0xDD: restart_interval = be16(seg); break;
			case 0xDA: { // SOS + entropy-coded data
				if (!have_frame) { why = "JPEG scan before frame header"; return false; }
				const int ns = d[seg];
				if (ns != (int)comps.size()) { why = "non-interleaved JPEG scans are not supported"; return false; }
				for (int i = 0; i < ns; ++i) {
					const int cid = d[seg + 1 + 2 * (size_t)i], tables = d[seg + 2 + 2 * (size_t)i];
					bool found = false;
					for (Component& c : comps)
						if (c.id == cid) { c.td = tables >> 4; c.ta = tables & 15; found = true; }
					if (!found) { why = "bad JPEG scan header"; return false; }
				}
				for (Component& c : comps)
					if (c.td > 3 || c.ta > 3 || !hdc[c.td].present || !hac[c.ta].present || !have_qt[c.tq]) { why = "JPEG scan refers to a missing table"; return false; }
				const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
				const int mcus_x = (W + mcu_w - 1) / mcu_w, mcus_y = (H + mcu_h - 1) / mcu_h;
				for (Component& c : comps) {
					c.w = mcus_x * c.h * 8;
					c.hgt = mcus_y * c.v * 8;
					c.plane.assign((size_t)c.w * c.hgt, 0);
					c.pred = 0;
				}
				BitReader br;
				br.p = d + seg_end;
				br.end = d + n;
				int until_restart = restart_interval;
				float block[64];
				for (int my = 0; my < mcus_y; ++my) {
					for (int mx = 0; mx < mcus_x; ++mx) {
						if (restart_interval && until_restart == 0) {
							// byte-align, expect RSTn
							br.reset();
							while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] >= 0xD0 && br.p[1] <= 0xD7)) ++br.p;
							if (br.p + 1 < br.end) br.p += 2;
							for (Component& c : comps) c.pred = 0;
							until_restart = restart_interval;
						}
						if (br.padded > 16) { why = "truncated JPEG data"; return false; } // the lookahead never needs more than a few bytes of padding
						for (Component& c : comps) {
							for (int by = 0; by < c.v; ++by) {
								for (int bx = 0; bx < c.h; ++bx) {
									memset(block, 0, sizeof(block));
									const int s = decode_symbol(br, hdc[c.td]);
									if (s < 0 || s > 11) { why = "corrupt JPEG data"; return false; }
									c.pred += extend(br.get(s), s);
									block[0] = (float)(c.pred * (int)qt[c.tq][0]);
									for (int k = 1; k < 64;) {
										const int rs = decode_symbol(br, hac[c.ta]);
										if (rs < 0) { why = "corrupt JPEG data"; return false; }
										const int r = rs >> 4, sz = rs & 15;
										if (sz == 0) {
											if (r != 15) break; // end of block
											k += 16;
											continue;
										}
										k += r;
										if (k > 63) { why = "corrupt JPEG data"; return false; }
										block[ZIGZAG[k]] = (float)(extend(br.get(sz), sz) * (int)qt[c.tq][k]);
										++k;
									}
									idct8x8(block, c.plane.data() + (size_t)(my * c.v + by) * 8 * c.w + (size_t)(mx * c.h + bx) * 8, c.w);
								}
							}
						}
						if (restart_interval) --until_restart;
					}
				}
				done = true;
				break;
			}
			default: break; // APPn, COM, ...
		}
		pos = seg_end;
	}
	if (!done) { why = "JPEG file without image data"; return false; }
	// full-resolution planes
	std::vector<std::vector<uint8_t>> full(comps.size());
	const int fw = ((W + 8 * hmax - 1) / (8 * hmax)) * 8 * hmax;
	for (size_t i = 0; i < comps.size(); ++i) {
		const Component& c = comps[i];
		if (hmax % c.h != 0 || vmax % c.v != 0) { why = "fractional JPEG sampling factors are not supported"; return false; }
		full[i] = upsample(c.plane, c.w, c.hgt, hmax / c.h, vmax / c.v);
	}
	rgba.resize((size_t)W * H * 4);
	for (int y = 0; y < H; ++y) {
		for (int x = 0; x < W; ++x) {
			uint8_t* o = rgba.data() + ((size_t)y * W + x) * 4;
			const size_t i = (size_t)y * fw + x;
			if (comps.size() == 1) {
				o[0] = o[1] = o[2] = full[0][i];
			} else {
				const float Y = full[0][i], cb = (float)full[1][i] - 128.0f, cr = (float)full[2][i] - 128.0f;
				const int r = (int)std::lrintf(Y + 1.402f * cr), g = (int)std::lrintf(Y - 0.344136f * cb - 0.714136f * cr), b = (int)std::lrintf(Y + 1.772f * cb);
				o[0] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
				o[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
				o[2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
			}
			o[3] = 255;
		}
	}
	width = W;
	height = H;
	return true;
}

} // namespace ngp
