// wavio.cpp -- the file half of the rosjack output stage and the batch front-end that goes with it.
//
// Reference: rosjack.cpp:189-210 opens `audio_file_path` with libsndfile as SF_FORMAT_WAV | SF_FORMAT_PCM_16, one channel,
// at the JACK (or resampled) rate; rosjack.cpp:404-409 copies every output period into write_file_buffer and calls
// sf_write_float(audio_file, write_file_buffer, data_length).  libsndfile is a system dependency of the reference (package.xml;
// Ubuntu 20.04 ships 1.0.28), absent from /root/reference and from this image, so its behaviour on this call path is restated:
//   * header: the canonical 44-byte RIFF/WAVE header of a WAVE_FORMAT_PCM file ('fmt ' chunk of 16 bytes, then 'data');
//     the two length fields are patched when the file is closed (sf_close);
//   * samples: sf_write_float on a PCM_16 file with the defaults norm_float = SF_TRUE and add_clipping = SF_FALSE converts
//     with f2s_array (src/pcm.c): dest = lrintf(src * 32767.0f) stored as short -- current rounding mode (nearest-even),
//     NO clipping: a sample beyond +-1.0 wraps modulo 2^16 exactly as the cast to short does there.
//   * reading (sf_read_float on PCM files, the front-end): PCM16 -> float by x / 32768 (s2f_array, norm 1/0x8000),
//     24- and 32-bit PCM by x / 2^23 and x / 2^31, IEEE float as is.
// The resampling half (rosjack.cpp:311-350, libsamplerate SRC_SINC_FASTEST) lives in resample.hip (bf_resampler_*).
//
// Host code only (file I/O).  The float -> PCM16 conversion of a batch that is resident in HBM has a device entry point
// (convert.hip) so that the D2H copy moves 2 bytes per sample instead of 4.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/bfcore.h"

struct bf_wav_writer {
    FILE *f = nullptr;
    uint64_t n_samples = 0;
    std::vector<int16_t> buf;
};

namespace {

void put_u16(unsigned char *p, unsigned v) { p[0] = v & 0xff; p[1] = (v >> 8) & 0xff; }
void put_u32(unsigned char *p, uint32_t v) { p[0] = v & 0xff; p[1] = (v >> 8) & 0xff; p[2] = (v >> 16) & 0xff; p[3] = (v >> 24) & 0xff; }
uint32_t get_u32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
unsigned get_u16(const unsigned char *p) { return p[0] | (p[1] << 8); }

void make_header(unsigned char (&h)[44], int sample_rate, uint32_t data_bytes) {
    memcpy(h, "RIFF", 4);
    put_u32(h + 4, 36 + data_bytes);
    memcpy(h + 8, "WAVEfmt ", 8);
    put_u32(h + 16, 16);                      // fmt chunk size
    put_u16(h + 20, 1);                       // WAVE_FORMAT_PCM
    put_u16(h + 22, 1);                       // channels (audio_info.channels = 1, rosjack.cpp:197)
    put_u32(h + 24, (uint32_t)sample_rate);
    put_u32(h + 28, (uint32_t)sample_rate * 2);  // bytes per second
    put_u16(h + 32, 2);                       // block align
    put_u16(h + 34, 16);                      // bits per sample
    memcpy(h + 36, "data", 4);
    put_u32(h + 40, data_bytes);
}

}  // namespace

extern "C" {

void bf_float_to_pcm16(const float *src, int16_t *dst, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const float scaled = src[i] * 32767.0f;           // float product, as src [count] * normfact in f2s_array
        const long r = lrintf(scaled);                    // nearest-even in the default rounding mode
        dst[i] = (int16_t)(uint16_t)(unsigned long)r;     // the cast to short: modulo 2^16, no clipping
    }
}

int bf_wav_writer_open(const char *path, int sample_rate, bf_wav_writer **out) {
    if (!path || !out || sample_rate <= 0) return BF_EINVAL;
    *out = nullptr;
    FILE *f = fopen(path, "wb");
    if (!f) return BF_ENOENT;
    unsigned char h[44];
    make_header(h, sample_rate, 0);
    if (fwrite(h, 1, 44, f) != 44) {
        fclose(f);
        return BF_EIO;
    }
    bf_wav_writer *w = new bf_wav_writer();
    w->f = f;
    *out = w;
    return BF_OK;
}

int bf_wav_writer_write(bf_wav_writer *w, const float *samples, size_t n) {
    if (!w || !w->f || (!samples && n)) return BF_EINVAL;
    w->buf.resize(n);
    bf_float_to_pcm16(samples, w->buf.data(), n);
    if (fwrite(w->buf.data(), 2, n, w->f) != n) return BF_EIO;   // little-endian host (x86-64), as the WAV file
    w->n_samples += n;
    return BF_OK;
}

int bf_wav_writer_write_pcm16(bf_wav_writer *w, const int16_t *pcm, size_t n) {
    if (!w || !w->f || (!pcm && n)) return BF_EINVAL;
    if (fwrite(pcm, 2, n, w->f) != n) return BF_EIO;
    w->n_samples += n;
    return BF_OK;
}

int bf_wav_writer_close(bf_wav_writer *w) {
    if (!w) return BF_EINVAL;
    int rc = BF_OK;
    if (w->f) {
        // patch the RIFF and data lengths, then close (what sf_close does for a file opened for writing)
        const uint64_t bytes = w->n_samples * 2;
        if (fflush(w->f) != 0) rc = BF_EIO;
        unsigned char len[4];
        put_u32(len, (uint32_t)(36 + bytes));
        if (rc == BF_OK && (fseek(w->f, 4, SEEK_SET) != 0 || fwrite(len, 1, 4, w->f) != 4)) rc = BF_EIO;
        put_u32(len, (uint32_t)bytes);
        if (rc == BF_OK && (fseek(w->f, 40, SEEK_SET) != 0 || fwrite(len, 1, 4, w->f) != 4)) rc = BF_EIO;
        if (fclose(w->f) != 0) rc = BF_EIO;
    }
    delete w;
    return rc;
}

// WAV reader: RIFF/WAVE, 'fmt ' PCM (1) 16/24/32 bit, IEEE float (3) 32 bit, or WAVE_FORMAT_EXTENSIBLE (0xFFFE) wrapping
// either; any channel count; unknown chunks are skipped.  Output: planar float32 [channel][sample] (what bf_process_batch
// takes as BF_PLANAR), malloc'ed -- release with bf_wav_free.
int bf_wav_read(const char *path, float **planar, int *n_channels, size_t *n_samples, int *sample_rate) {
    if (!path || !planar || !n_channels || !n_samples) return BF_EINVAL;
    *planar = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return BF_ENOENT;
    // chunk lengths come from the file: never allocate more than the file can hold (a corrupt or streamed header says 4 GiB)
    long file_bytes = -1;
    if (fseek(f, 0, SEEK_END) == 0) file_bytes = ftell(f);
    if (file_bytes < 0 || fseek(f, 0, SEEK_SET) != 0) {
        fclose(f);
        return BF_EIO;
    }
    try {  // std::bad_alloc must not cross the extern "C" boundary
    unsigned char hd[12];
    int rc = BF_EINVAL;
    unsigned fmt = 0, ch = 0, bits = 0;
    uint32_t rate = 0;
    std::vector<unsigned char> data;
    if (fread(hd, 1, 12, f) == 12 && memcmp(hd, "RIFF", 4) == 0 && memcmp(hd + 8, "WAVE", 4) == 0) {
        bool have_fmt = false, have_data = false;
        unsigned char ck[8];
        while (!have_data && fread(ck, 1, 8, f) == 8) {
            const uint32_t sz = get_u32(ck + 4);
            const long here = ftell(f);
            const uint64_t left = here >= 0 && here <= file_bytes ? (uint64_t)(file_bytes - here) : 0;
            if (memcmp(ck, "fmt ", 4) == 0 && sz >= 16) {
                if (sz > 4096 || sz > left) break;  // WAVEFORMATEXTENSIBLE is 40 bytes
                std::vector<unsigned char> b(sz);
                if (fread(b.data(), 1, sz, f) != sz) break;
                fmt = get_u16(b.data());
                ch = get_u16(b.data() + 2);
                rate = get_u32(b.data() + 4);
                bits = get_u16(b.data() + 14);
                if (fmt == 0xFFFE && sz >= 26) fmt = get_u16(b.data() + 24);  // sub-format GUID's first two bytes
                have_fmt = true;
                if (sz & 1) fseek(f, 1, SEEK_CUR);
            } else if (memcmp(ck, "data", 4) == 0) {
                if (!have_fmt) break;
                const size_t want = sz < left ? (size_t)sz : (size_t)left;
                data.resize(want);
                const size_t got = want ? fread(data.data(), 1, want, f) : 0;
                data.resize(got);  // a truncated or still-open file (length fields 0): take what is there
                if (sz == 0) {
                    unsigned char tmp[65536];
                    size_t k;
                    while ((k = fread(tmp, 1, sizeof(tmp), f)) > 0) data.insert(data.end(), tmp, tmp + k);
                }
                have_data = true;
            } else {
                if (fseek(f, (long)sz + (sz & 1), SEEK_CUR) != 0) break;
            }
        }
        const bool pcm = fmt == 1 && (bits == 16 || bits == 24 || bits == 32), flt = fmt == 3 && bits == 32;
        if (have_fmt && have_data && ch >= 1 && (pcm || flt)) {
            const size_t bps = bits / 8, frames = data.size() / (bps * ch);
            float *out = (float *)malloc(sizeof(float) * (frames * ch ? frames * ch : 1));
            if (!out) {
                rc = BF_ENOMEM;
            } else {
                for (size_t n = 0; n < frames; ++n)
                    for (unsigned c = 0; c < ch; ++c) {
                        const unsigned char *p = data.data() + (n * ch + c) * bps;
                        float v;
                        if (flt) {
                            memcpy(&v, p, 4);
                        } else if (bits == 16) {
                            v = (float)(int16_t)get_u16(p) * (1.0f / 32768.0f);
                        } else if (bits == 24) {
                            int32_t s = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24) >> 8;
                            v = (float)s * (1.0f / 8388608.0f);
                        } else {
                            v = (float)(int32_t)get_u32(p) * (1.0f / 2147483648.0f);
                        }
                        out[(size_t)c * frames + n] = v;
                    }
                *planar = out;
                *n_channels = (int)ch;
                *n_samples = frames;
                if (sample_rate) *sample_rate = (int)rate;
                rc = BF_OK;
            }
        }
    }
    fclose(f);
    return rc;
    } catch (...) {
        fclose(f);
        return BF_ENOMEM;
    }
}

// Raw planar float32 file: [n_channels][n_samples] little-endian floats, nothing else (the layout bf_process_batch takes).
int bf_planar_f32_read(const char *path, int n_channels, float **planar, size_t *n_samples) {
    if (!path || !planar || !n_samples || n_channels < 1) return BF_EINVAL;
    *planar = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return BF_ENOENT;
    int rc = BF_EINVAL;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long bytes = ftell(f);
        if (bytes >= 0 && bytes % (4L * n_channels) == 0 && fseek(f, 0, SEEK_SET) == 0) {
            float *out = (float *)malloc(bytes > 0 ? (size_t)bytes : 4);
            if (!out) {
                rc = BF_ENOMEM;
            } else if (fread(out, 1, (size_t)bytes, f) != (size_t)bytes) {
                free(out);
                rc = BF_EIO;
            } else {
                *planar = out;
                *n_samples = (size_t)bytes / 4 / (size_t)n_channels;
                rc = BF_OK;
            }
        }
    }
    fclose(f);
    return rc;
}

void bf_wav_free(float *planar) { free(planar); }

}  // extern "C"
