// RGB8 -> PNG file bytes on the host: the last step of run_job (the reference's ``img.save(buf, format="PNG")``,
// backends/cuda_worker.py:234-239).  With the sampler at ~20 ms a general-purpose deflate is the largest host cost of a
// request (zlib level 1: 16 ms of CPU for 512x512, 3 ms on 8 threads), so the stream is written directly:
//   * scanline filter 2 ("Up"), as before;
//   * the image is cut into stripes of whole scanlines; every stripe becomes ONE dynamic-Huffman deflate block over
//     literals plus distance-1 runs (length 3..258: flat areas filter to runs of zeros) -- no hash chains, no window;
//     a stripe whose Huffman form would be larger than its bytes is written as stored blocks;
//   * stripes end on a byte boundary (an empty stored block), are compressed on a small pool of threads and concatenated
//     behind one zlib header; Adler-32 and the IDAT CRC-32 are computed per stripe and combined.
// Lossless and deterministic: the bytes depend on (image, stripe count) only.  Any PNG reader decodes the result.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define LCM_OK 0
#define LCM_EINVAL (-1)
void lcm_set_error(const char* fmt, ...);

namespace {

// ------------------------------------------------------------------------------------------------ checksums
uint32_t g_crc_tab[8][256];
std::once_flag g_crc_once;
void crc_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        g_crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = g_crc_tab[0][g_crc_tab[t - 1][i] & 0xFF] ^ (g_crc_tab[t - 1][i] >> 8);
}
// CRC-32 (IEEE) of buf continuing from `crc` (0 to start), slicing by 8
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    std::call_once(g_crc_once, crc_init);
    uint32_t c = ~crc;
    while (n && ((uintptr_t)p & 7)) { c = g_crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8); --n; }
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        const uint32_t lo = (uint32_t)v ^ c, hi = (uint32_t)(v >> 32);
        c = g_crc_tab[7][lo & 0xFF] ^ g_crc_tab[6][(lo >> 8) & 0xFF] ^ g_crc_tab[5][(lo >> 16) & 0xFF] ^ g_crc_tab[4][lo >> 24] ^
            g_crc_tab[3][hi & 0xFF] ^ g_crc_tab[2][(hi >> 8) & 0xFF] ^ g_crc_tab[1][(hi >> 16) & 0xFF] ^ g_crc_tab[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = g_crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}
// CRC of A||B from crc(A), crc(B), len(B): multiply crc(A) by x^(8 len(B)) in GF(2)[x] / P (square-and-multiply on 32x32 bit matrices)
uint32_t gf2_times(const uint32_t* m, uint32_t v) {
    uint32_t s = 0;
    for (; v; v >>= 1, ++m)
        if (v & 1) s ^= *m;
    return s;
}
void gf2_square(uint32_t* sq, const uint32_t* m) {
    for (int i = 0; i < 32; ++i) sq[i] = gf2_times(m, m[i]);
}
uint32_t crc32_concat(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    if (len_b == 0) return crc_a;
    uint32_t even[32], odd[32];
    odd[0] = 0xEDB88320u;                                  // operator for one zero BIT
    for (int i = 1; i < 32; ++i) odd[i] = 1u << (i - 1);
    gf2_square(even, odd);                                 // two bits
    gf2_square(odd, even);                                 // four bits
    do {
        gf2_square(even, odd);                             // first pass: one zero byte
        if (len_b & 1) crc_a = gf2_times(even, crc_a);
        len_b >>= 1;
        if (!len_b) break;
        gf2_square(odd, even);
        if (len_b & 1) crc_a = gf2_times(odd, crc_a);
        len_b >>= 1;
    } while (len_b);
    return crc_a ^ crc_b;
}

constexpr uint32_t ADLER_BASE = 65521u;
uint32_t adler32_update(uint32_t adler, const uint8_t* p, size_t n) {
    uint32_t a = adler & 0xFFFF, b = adler >> 16;
    while (n) {
        size_t k = n < 5552 ? n : 5552;                    // largest run before b can overflow 32 bits
        n -= k;
        while (k >= 8) {
            a += p[0]; b += a; a += p[1]; b += a; a += p[2]; b += a; a += p[3]; b += a;
            a += p[4]; b += a; a += p[5]; b += a; a += p[6]; b += a; a += p[7]; b += a;
            p += 8; k -= 8;
        }
        while (k--) { a += *p++; b += a; }
        a %= ADLER_BASE; b %= ADLER_BASE;
    }
    return (b << 16) | a;
}
// Adler-32 of A||B from adler(A), adler(B) (both started at 1) and len(B)
uint32_t adler32_concat(uint32_t ad_a, uint32_t ad_b, uint64_t len_b) {
    const uint32_t rem = (uint32_t)(len_b % ADLER_BASE);
    uint32_t s1 = ad_a & 0xFFFF;
    uint32_t s2 = (uint32_t)(((uint64_t)rem * s1) % ADLER_BASE);
    s1 += (ad_b & 0xFFFF) + ADLER_BASE - 1;
    s2 += (ad_a >> 16) + (ad_b >> 16) + ADLER_BASE - rem;
    if (s1 >= ADLER_BASE) s1 -= ADLER_BASE;
    if (s1 >= ADLER_BASE) s1 -= ADLER_BASE;
    if (s2 >= (ADLER_BASE << 1)) s2 -= (ADLER_BASE << 1);
    if (s2 >= ADLER_BASE) s2 -= ADLER_BASE;
    return (s2 << 16) | s1;
}

// ------------------------------------------------------------------------------------------------ Huffman code lengths
// Optimal code lengths by the two-queue merge over the frequency-sorted symbols, then limited to `maxlen` by moving the
// overlong codes up and paying for them with the shortest-possible demotions (the Kraft sum is restored exactly), and
// handed back to the symbols in frequency order.  Symbols with frequency 0 get length 0; a lone used symbol gets length 1.
void huff_lengths(const uint32_t* freq, int n, int maxlen, uint8_t* len) {
    struct Node { uint64_t w; int l, r; };
    int order[288], used = 0;
    for (int i = 0; i < n; ++i) {
        len[i] = 0;
        if (freq[i]) order[used++] = i;
    }
    if (used == 0) return;
    if (used == 1) { len[order[0]] = 1; return; }
    std::sort(order, order + used, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    Node nodes[2 * 288];
    for (int i = 0; i < used; ++i) nodes[i] = {freq[order[i]], -1, -1};
    int leaf = 0, inner = used, made = used;
    auto take = [&]() {
        if (leaf < used && (inner >= made || nodes[leaf].w <= nodes[inner].w)) return leaf++;
        return inner++;
    };
    while ((used - leaf) + (made - inner) > 1) {
        const int a = take(), b = take();
        nodes[made] = {nodes[a].w + nodes[b].w, a, b};
        ++made;
    }
    // depth of every leaf: parents come after their children, so walk from the root down
    int depth[2 * 288];
    depth[made - 1] = 0;
    int count[64] = {0};
    for (int i = made - 1; i >= used; --i) {
        depth[nodes[i].l] = depth[i] + 1;
        depth[nodes[i].r] = depth[i] + 1;
    }
    for (int i = 0; i < used; ++i) count[depth[i] < 63 ? depth[i] : 63]++;
    // limit: fold everything deeper than maxlen into maxlen, then repair the Kraft sum
    for (int d = maxlen + 1; d < 64; ++d) { count[maxlen] += count[d]; count[d] = 0; }
    uint64_t total = 0;
    for (int d = maxlen; d >= 1; --d) total += (uint64_t)count[d] << (maxlen - d);
    while (total > (1ull << maxlen)) {
        count[maxlen]--;
        for (int d = maxlen - 1; d >= 1; --d)
            if (count[d]) { count[d]--; count[d + 1] += 2; break; }
        total--;
    }
    // the rarest symbols take the longest codes
    int k = 0;
    for (int d = maxlen; d >= 1; --d)
        for (int c = count[d]; c > 0; --c) len[order[k++]] = (uint8_t)d;
}

// canonical codes (RFC 1951 3.2.2), bit-reversed for the LSB-first stream
void huff_codes(const uint8_t* len, int n, uint16_t* code) {
    int bl_count[16] = {0}, next[16];
    for (int i = 0; i < n; ++i) bl_count[len[i]]++;
    bl_count[0] = 0;
    int c = 0;
    for (int b = 1; b <= 15; ++b) { c = (c + bl_count[b - 1]) << 1; next[b] = c; }
    for (int i = 0; i < n; ++i) {
        if (!len[i]) { code[i] = 0; continue; }
        uint32_t v = (uint32_t)next[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; ++b) { r = (r << 1) | (v & 1); v >>= 1; }
        code[i] = (uint16_t)r;
    }
}

struct BitWriter {
    uint8_t* p;
    uint8_t* end;
    uint64_t acc = 0;
    int nbits = 0;
    bool overflow = false;
    inline void put(uint32_t v, int n) {           // n <= 32
        acc |= (uint64_t)v << nbits;
        nbits += n;
        if (nbits >= 32) {
            if (p + 4 > end) { overflow = true; nbits -= 32; acc >>= 32; return; }
            const uint32_t w = (uint32_t)acc;
            memcpy(p, &w, 4);
            p += 4; acc >>= 32; nbits -= 32;
        }
    }
    inline void align() {
        while (nbits > 0) {
            if (p >= end) { overflow = true; break; }
            *p++ = (uint8_t)acc; acc >>= 8; nbits -= 8;
        }
        acc = 0; nbits = 0;
    }
    inline void bytes(const uint8_t* s, size_t n) {   // after align()
        if (p + n > end) { overflow = true; return; }
        memcpy(p, s, n); p += n;
    }
};

// length 3..258 -> (code 257.., extra bits, extra value)
const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
uint8_t g_len_sym[259];
std::once_flag g_len_once;
void len_init() {
    for (int l = 3; l <= 258; ++l) {
        int s = 28;
        while (LEN_BASE[s] > l) --s;
        g_len_sym[l] = (uint8_t)s;
    }
}

struct Stripe {
    const uint8_t* rgb;       // first scanline of the stripe
    const uint8_t* above;     // the scanline above it, or null for the top of the image
    long long pitch;          // bytes between scanlines of the source
    int rows, row_bytes;      // row_bytes = 3 * width
    bool last;
    uint8_t* out;             // raw deflate segment goes here (a slice of the caller's buffer, out_cap bytes)
    size_t out_cap;
    size_t out_len = 0;
    uint32_t adler = 1, crc = 0;
    uint64_t raw_len = 0;
    bool failed = false;
};

inline size_t stripe_bound(size_t raw) { return raw + 5 * ((raw + 65534) / 65535) + 5 + 16; }

// Matches of the filtered bytes, in order: `len` (3..258) bytes repeating at distance 1 (a run) or 3 (the same channel of the
// previous pixel: smooth vertical gradients filter to a constant per channel); the longer wins.  Everything between two
// matches is literals.  A match is stored as pos | then len | dcode << 16 (distance code 0 = distance 1, 2 = distance 3).
inline uint32_t load32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
size_t find_matches(const uint8_t* f, size_t n, std::vector<uint32_t>& m) {
    size_t nm = 0;
    if (m.size() < 2 * (n / 3 + 2)) m.resize(2 * (n / 3 + 2));
    uint32_t* o = m.data();
    size_t i = 3;
    const size_t stop = n >= 8 ? n - 8 : 0;          // the 32-bit probes stay inside the buffer; the tail goes out as literals
    while (i < stop) {
        const uint32_t cur = load32(f + i), prev1 = load32(f + i - 1), prev3 = load32(f + i - 3);
        const bool run = ((cur ^ prev1) & 0xFFFFFF) == 0;             // f[i..i+2] == f[i-1..i+1]: three more of the same byte
        const bool rep = ((cur ^ prev3) & 0xFFFFFF) == 0;             // f[i..i+2] == f[i-3..i-1]
        if (!(run | rep)) { ++i; continue; }
        const size_t lim = std::min(n, i + 258);
        const int d = run ? 1 : 3;                                    // a run is also a distance-3 repeat once it is 3 long: try 1 first
        size_t j = i + 3;
        while (j + 8 <= lim && load64(f + j) == load64(f + j - d)) j += 8;
        while (j < lim && f[j] == f[j - d]) ++j;
        size_t l = j - i;
        int dc = run ? 0 : 2;
        if (run && rep && l < 258) {                                  // both start here: keep the longer
            size_t k = i + 3;
            while (k + 8 <= lim && load64(f + k) == load64(f + k - 3)) k += 8;
            while (k < lim && f[k] == f[k - 3]) ++k;
            if (k - i > l) { l = k - i; dc = 2; }
        }
        o[nm++] = (uint32_t)i;
        o[nm++] = (uint32_t)l | ((uint32_t)dc << 16);
        i += l;
    }
    return nm / 2;
}

inline void histogram(const uint8_t* p, size_t n, uint32_t* freq) {
    uint32_t h[4][256];
    memset(h, 0, sizeof(h));
    size_t i = 0;
    for (; i + 4 <= n; i += 4) { h[0][p[i]]++; h[1][p[i + 1]]++; h[2][p[i + 2]]++; h[3][p[i + 3]]++; }
    for (; i < n; ++i) h[0][p[i]]++;
    for (int s = 0; s < 256; ++s) freq[s] += h[0][s] + h[1][s] + h[2][s] + h[3][s];
}

// One stripe -> filtered scanlines -> one dynamic block (or stored blocks) + byte alignment.  Two passes over the filtered
// bytes (histogram, then emit); the scratch lives with the thread, so a call allocates nothing after the first.
void encode_stripe(Stripe& st) {
    std::call_once(g_len_once, len_init);
    const int rb = st.row_bytes, line = rb + 1;
    const size_t n = (size_t)line * st.rows;
    st.raw_len = n;
    static thread_local std::vector<uint8_t> scratch;
    if (scratch.size() < n) scratch.resize(n);
    uint8_t* filt = scratch.data();
    for (int y = 0; y < st.rows; ++y) {
        uint8_t* d = filt + (size_t)y * line;
        const uint8_t* cur = st.rgb + (long long)y * st.pitch;
        const uint8_t* up = y ? cur - st.pitch : st.above;
        d[0] = 2;                                    // filter type Up
        if (up) {
            for (int i = 0; i < rb; ++i) d[1 + i] = (uint8_t)(cur[i] - up[i]);
        } else {
            memcpy(d + 1, cur, rb);
        }
    }
    st.adler = adler32_update(1, filt, n);

    static thread_local std::vector<uint32_t> mbuf;
    const size_t nm = find_matches(filt, n, mbuf);
    const uint32_t* mt = mbuf.data();
    uint32_t freq[288] = {0}, dfreq[4] = {0};
    if (nm == 0) {
        histogram(filt, n, freq);
    } else {
        size_t at = 0;
        for (size_t k = 0; k < nm; ++k) {
            const size_t pos = mt[2 * k], l = mt[2 * k + 1] & 0xFFFF;
            if (pos - at >= 64) histogram(filt + at, pos - at, freq);
            else for (size_t i = at; i < pos; ++i) freq[filt[i]]++;
            freq[257 + g_len_sym[l]]++;
            dfreq[mt[2 * k + 1] >> 16]++;
            at = pos + l;
        }
        if (n - at >= 64) histogram(filt + at, n - at, freq);
        else for (size_t i = at; i < n; ++i) freq[filt[i]]++;
    }
    freq[256] = 1;                                   // end of block
    uint8_t ll_len[288], d_len[4];
    uint16_t ll_code[288], d_code[4];
    huff_lengths(freq, 286, 15, ll_len);
    huff_codes(ll_len, 286, ll_code);
    // distance codes 0 (distance 1) and 2 (distance 3); always a complete tree of >= 2 codes, which every inflater accepts
    if (!dfreq[0]) dfreq[0] = 1;
    if (!dfreq[2]) dfreq[2] = 1;
    huff_lengths(dfreq, 3, 15, d_len);
    huff_codes(d_len, 3, d_code);
    // code lengths, sent plainly (no repeat codes): the header is ~150 bytes per stripe either way
    int hlit = 286;
    while (hlit > 257 && ll_len[hlit - 1] == 0) --hlit;
    uint8_t seq[288 + 4];
    int nseq = 0;
    for (int i = 0; i < hlit; ++i) seq[nseq++] = ll_len[i];
    for (int i = 0; i < 3; ++i) seq[nseq++] = d_len[i];
    uint32_t cl_freq[19] = {0};
    for (int i = 0; i < nseq; ++i) cl_freq[seq[i]]++;
    uint8_t cl_len[19];
    uint16_t cl_code[19];
    huff_lengths(cl_freq, 19, 7, cl_len);
    huff_codes(cl_len, 19, cl_code);
    static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cl_len[CL_ORDER[hclen - 1]] == 0) --hclen;

    uint64_t bits = 3 + 5 + 5 + 4 + 3ull * hclen;
    for (int i = 0; i < nseq; ++i) bits += cl_len[seq[i]];
    for (int s = 0; s < 286; ++s) bits += (uint64_t)freq[s] * ll_len[s];
    for (int s = 257; s < 286; ++s) bits += (uint64_t)freq[s] * (LEN_EXTRA[s - 257] + 1);      // extra bits + <= 1 distance bit
    const size_t stored_bytes = n + 5 * ((n + 65534) / 65535);
    const bool stored = (bits + 7) / 8 + 8 >= stored_bytes;

    BitWriter bw{st.out, st.out + st.out_cap};
    if (stored) {
        size_t off = 0;
        while (off < n) {
            const size_t k = std::min<size_t>(65535, n - off);
            bw.put(0, 3);                            // BFINAL 0, BTYPE 00
            bw.align();
            const uint8_t hdr[4] = {(uint8_t)(k & 0xFF), (uint8_t)(k >> 8), (uint8_t)(~k & 0xFF), (uint8_t)((~k >> 8) & 0xFF)};
            bw.bytes(hdr, 4);
            bw.bytes(filt + off, k);
            off += k;
        }
    } else {
        bw.put(0, 1);                                // BFINAL 0 (the closing block follows)
        bw.put(2, 2);                                // dynamic Huffman
        bw.put((uint32_t)(hlit - 257), 5);
        bw.put(2, 5);                                // HDIST: 3 distance codes
        bw.put((uint32_t)(hclen - 4), 4);
        for (int i = 0; i < hclen; ++i) bw.put(cl_len[CL_ORDER[i]], 3);
        for (int i = 0; i < nseq; ++i) bw.put(cl_code[seq[i]], cl_len[seq[i]]);
        // symbol -> (code | length << 16) once, then a literal is a table look-up and a shift
        uint32_t lit[256];
        for (int s = 0; s < 256; ++s) lit[s] = (uint32_t)ll_code[s] | ((uint32_t)ll_len[s] << 16);
        auto literals = [&](size_t from, size_t to) {
            for (size_t i = from; i < to; ++i) { const uint32_t e = lit[filt[i]]; bw.put(e & 0xFFFF, (int)(e >> 16)); }
        };
        size_t at = 0;
        for (size_t k = 0; k < nm; ++k) {
            const size_t pos = mt[2 * k];
            const int l = (int)(mt[2 * k + 1] & 0xFFFF), dc = (int)(mt[2 * k + 1] >> 16), sy = g_len_sym[l];
            literals(at, pos);
            bw.put(ll_code[257 + sy], ll_len[257 + sy]);
            if (LEN_EXTRA[sy]) bw.put((uint32_t)(l - LEN_BASE[sy]), LEN_EXTRA[sy]);
            bw.put(d_code[dc], d_len[dc]);           // distances 1 and 3 carry no extra bits
            at = pos + l;
        }
        literals(at, n);
        bw.put(ll_code[256], ll_len[256]);
    }
    // close on a byte boundary: an empty stored block, final on the last stripe
    bw.put(st.last ? 1 : 0, 3);
    bw.align();
    const uint8_t empty[4] = {0, 0, 0xFF, 0xFF};
    bw.bytes(empty, 4);
    st.failed = bw.overflow;
    st.out_len = (size_t)(bw.p - st.out);
    st.crc = crc32_update(0, st.out, st.out_len);
}

// ------------------------------------------------------------------------------------------------ worker threads
// A handful of detached threads, started on first use; a call hands them its stripes and works on them itself as well.
class Pool {
  public:
    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) std::thread([this] { run(); }).detach();
    }
    void for_each(std::vector<Stripe>& items) {
        struct Batch { size_t next = 0, done = 0; } b;
        {
            std::lock_guard<std::mutex> g(m_);
            owners_.push_back({&b.next, &b.done, &items});
        }
        cv_.notify_all();
        // the caller takes items too (no idle wait, and progress is guaranteed whatever the pool is busy with)
        for (;;) {
            size_t i;
            {
                std::lock_guard<std::mutex> g(m_);
                if (b.next >= items.size()) break;
                i = b.next++;
            }
            encode_stripe(items[i]);
            std::lock_guard<std::mutex> g(m_);
            b.done++;
        }
        std::unique_lock<std::mutex> g(m_);
        done_cv_.wait(g, [&] { return b.done == items.size(); });
        for (size_t k = 0; k < owners_.size(); ++k)
            if (owners_[k].next == &b.next) { owners_.erase(owners_.begin() + k); break; }
    }

  private:
    struct Owner { size_t* next; size_t* done; std::vector<Stripe>* items; };
    void run() {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            Owner* o = nullptr;
            for (auto& c : owners_)
                if (*c.next < c.items->size()) { o = &c; break; }
            if (!o) { cv_.wait(g); continue; }
            const size_t i = (*o->next)++;
            std::vector<Stripe>* items = o->items;
            size_t* done = o->done;
            g.unlock();
            encode_stripe((*items)[i]);
            g.lock();
            (*done)++;
            done_cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_cv_;
    std::vector<Owner> owners_;
};

Pool* pool() {
    static Pool* p = [] {
        int n = 7;
        if (const char* e = getenv("LCM_PNG_POOL")) n = atoi(e);
        const unsigned hw = std::thread::hardware_concurrency();
        if (hw && n > (int)hw - 1) n = (int)hw - 1;
        if (n < 0) n = 0;
        return new Pool(n);                          // never destroyed: its threads are detached
    }();
    return p;
}

inline void put_be32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

}  // namespace

// Upper bound of the file size for lcm_png_encode_rgb8 (stored blocks + framing).
extern "C" long long lcm_png_bound(int width, int height, int stripes) {
    if (width <= 0 || height <= 0) return 0;
    if (stripes < 1) stripes = 1;
    const long long raw = ((long long)width * 3 + 1) * height;
    return raw + 5 * (raw / 65535 + stripes) + 32ll * stripes + 128;
}

// rgb: height scanlines of width RGB8 pixels, `pitch` bytes apart (>= 3*width).  stripes: how many deflate segments (and
// units of parallel work) the image is cut into; the bytes written depend on it, on nothing else.
extern "C" int lcm_png_encode_rgb8(const void* rgb, int width, int height, long long pitch, int stripes, void* out,
                                   long long out_cap, long long* out_len) {
    if (!rgb || !out || !out_len) { lcm_set_error("png_encode: null pointer"); return LCM_EINVAL; }
    if (width <= 0 || height <= 0 || pitch < 3ll * width) { lcm_set_error("png_encode: bad shape %dx%d pitch %lld", width, height, pitch); return LCM_EINVAL; }
    if (stripes < 1) stripes = 1;
    if (stripes > height) stripes = height;
    if (stripes > 64) stripes = 64;
    if (out_cap < lcm_png_bound(width, height, stripes)) { lcm_set_error("png_encode: output buffer %lld < bound %lld", out_cap, lcm_png_bound(width, height, stripes)); return LCM_EINVAL; }
    std::vector<Stripe> st(stripes);
    // every stripe writes into its own worst-case slice of the caller's buffer; the slices are then moved down into place
    uint8_t* const zbase = (uint8_t*)out + 8 + 25 + 10;
    size_t off = 0;
    for (int i = 0; i < stripes; ++i) {
        const int r0 = (int)((long long)height * i / stripes), r1 = (int)((long long)height * (i + 1) / stripes);
        st[i].rgb = (const uint8_t*)rgb + (long long)r0 * pitch;
        st[i].above = r0 ? st[i].rgb - pitch : nullptr;
        st[i].pitch = pitch;
        st[i].rows = r1 - r0;
        st[i].row_bytes = 3 * width;
        st[i].last = i == stripes - 1;
        st[i].out = zbase + off;
        st[i].out_cap = stripe_bound((size_t)(3 * width + 1) * (r1 - r0));
        off += st[i].out_cap;
    }
    if ((long long)(8 + 25 + 10 + off + 8 + 12) > out_cap) { lcm_set_error("png_encode: internal bound error"); return LCM_EINVAL; }
    if (stripes == 1) encode_stripe(st[0]);
    else pool()->for_each(st);

    uint8_t* o = (uint8_t*)out;
    static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    memcpy(o, SIG, 8); o += 8;
    {   // IHDR
        uint8_t c[4 + 13] = {'I', 'H', 'D', 'R'};
        put_be32(c + 4, (uint32_t)width); put_be32(c + 8, (uint32_t)height);
        c[12] = 8; c[13] = 2; c[14] = 0; c[15] = 0; c[16] = 0;
        put_be32(o, 13); memcpy(o + 4, c, 17); put_be32(o + 21, crc32_update(0, c, 17));
        o += 25;
    }
    size_t zlen = 2 + 4;
    for (auto& s : st) {
        if (s.failed) { lcm_set_error("png_encode: stripe buffer overflow"); return LCM_EINVAL; }
        zlen += s.out_len;
    }
    put_be32(o, (uint32_t)zlen);
    const uint8_t head[6] = {'I', 'D', 'A', 'T', 0x78, 0x01};
    memcpy(o + 4, head, 6);
    uint32_t crc = crc32_update(0, head, 6), adler = 1;
    uint8_t* z = o + 10;
    bool first = true;
    for (auto& s : st) {
        if (z != s.out) memmove(z, s.out, s.out_len);
        z += s.out_len;
        crc = crc32_concat(crc, s.crc, s.out_len);
        adler = first ? s.adler : adler32_concat(adler, s.adler, s.raw_len);
        first = false;
    }
    put_be32(z, adler);
    crc = crc32_concat(crc, crc32_update(0, z, 4), 4);
    put_be32(z + 4, crc);
    o = z + 8;
    static const uint8_t IEND[12] = {0, 0, 0, 0, 'I', 'E', 'N', 'D', 0xAE, 0x42, 0x60, 0x82};
    memcpy(o, IEND, 12); o += 12;
    *out_len = (long long)(o - (uint8_t*)out);
    return LCM_OK;
}
