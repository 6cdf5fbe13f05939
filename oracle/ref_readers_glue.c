/* TEST INFRASTRUCTURE ONLY -- a thin C driver around the REFERENCE's own native block readers
 * (fastcard/card_reader.c, fastcard/raw_reader.c, fastcard/lib/base64.c), which oracle/Makefile
 * compiles from where they lie under /root/reference into oracle/_ref/libfastcard_readers.so.
 * Nothing of the reference is copied here: this file only allocates the structs those readers
 * fill (reader.h: block_t, reader_settings_t) and forwards to the reader's own `next`.
 * (fastcard/reader.c itself -- reader_block_new() -- cannot be built here: it includes rawconv.h ->
 * fft.h -> <fftw3.h>, which this image lacks; so the block is allocated below, and filled with the
 * caller's initial bytes instead of reader.c's RAWCONV_ZERO.)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "card_reader.h"
#include "raw_reader.h"

typedef struct {
    reader_t* reader;
    block_t block;
    FILE* file;
    size_t block_size;
} ref_reader;

static ref_reader* ref_open(const char* path, size_t block_size, size_t history_size, int card,
                            const unsigned char* initial /* 2*block_size bytes or NULL */) {
    ref_reader* r = calloc(1, sizeof(ref_reader));
    if (!r) return NULL;
    r->file = fopen(path, "rb");
    if (!r->file) {
        free(r);
        return NULL;
    }
    r->block_size = block_size;
    r->block.raw_samples = malloc(block_size * sizeof(uint16_t) + 5);   /* (+5: as reader_block_new) */
    r->block.index = -1;
    if (initial)
        memcpy(r->block.raw_samples, initial, 2 * block_size);
    else
        memset(r->block.raw_samples, 0, 2 * block_size);
    reader_settings_t st;
    st.output = &r->block;
    st.block_size = block_size;
    st.history_size = history_size;
    r->reader = card ? card_reader_new(st, r->file) : raw_reader_new(st, r->file);
    if (!r->reader) {
        fclose(r->file);
        free(r->block.raw_samples);
        free(r);
        return NULL;
    }
    return r;
}

void* ref_card_open(const char* path, size_t block_size, size_t history_size, const unsigned char* initial) {
    return ref_open(path, block_size, history_size, 1, initial);
}

void* ref_raw_open(const char* path, size_t block_size, size_t history_size, const unsigned char* initial) {
    return ref_open(path, block_size, history_size, 0, initial);
}

/* the reader's own return code (0 block, 1 end of file, < 0 error); on 0 the block's bytes,
 * timestamp and index are copied out */
int ref_next(void* handle, unsigned char* out_bytes, long* tv_sec, long* tv_usec, long long* index) {
    ref_reader* r = handle;
    const int rc = r->reader->next(r->reader->context);
    if (rc == 0) {
        memcpy(out_bytes, r->block.raw_samples, 2 * r->block_size);
        *tv_sec = (long)r->block.timestamp.tv_sec;
        *tv_usec = (long)r->block.timestamp.tv_usec;
        *index = (long long)r->block.index;
    }
    return rc;
}

void ref_close(void* handle) {
    ref_reader* r = handle;
    if (!r) return;
    if (r->reader) {
        if (r->reader->free) r->reader->free(r->reader->context);
        free(r->reader);
    }
    if (r->file) fclose(r->file);
    free(r->block.raw_samples);
    free(r);
}
