/* TEST INFRASTRUCTURE (oracle).  Minimal reader for RRLWBLOB files (format: rrtmg_lw_amd/blob.py). */
#ifndef RRLW_BLOB_H
#define RRLW_BLOB_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    char name[48];
    uint32_t dtype, ndim, dims[6];
    uint64_t off, nbytes;
} rrlw_blob_entry;

typedef struct {
    unsigned char *buf;
    size_t size;
    uint32_t n;
    rrlw_blob_entry *ent;
} rrlw_blob;

static int rrlw_blob_open(rrlw_blob *b, const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    b->buf = (unsigned char *)malloc((size_t)sz);
    if (!b->buf || fread(b->buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return -2; }
    fclose(f);
    b->size = (size_t)sz;
    if (sz < 16 || memcmp(b->buf, "RRLWBLOB", 8) != 0) return -3;
    uint32_t ver;
    memcpy(&ver, b->buf + 8, 4);
    memcpy(&b->n, b->buf + 12, 4);
    if (ver != 1) return -4;
    b->ent = (rrlw_blob_entry *)malloc(sizeof(rrlw_blob_entry) * b->n);
    for (uint32_t i = 0; i < b->n; i++) {
        const unsigned char *p = b->buf + 16 + 96 * (size_t)i;
        memcpy(b->ent[i].name, p, 48);
        b->ent[i].name[47] = 0;
        memcpy(&b->ent[i].dtype, p + 48, 4);
        memcpy(&b->ent[i].ndim, p + 52, 4);
        memcpy(b->ent[i].dims, p + 56, 24);
        memcpy(&b->ent[i].off, p + 80, 8);
        memcpy(&b->ent[i].nbytes, p + 88, 8);
    }
    return 0;
}

static const rrlw_blob_entry *rrlw_blob_find(const rrlw_blob *b, const char *name)
{
    for (uint32_t i = 0; i < b->n; i++)
        if (strcmp(b->ent[i].name, name) == 0) return &b->ent[i];
    return NULL;
}

static const void *rrlw_blob_data(const rrlw_blob *b, const rrlw_blob_entry *e) { return b->buf + e->off; }

static void rrlw_blob_close(rrlw_blob *b)
{
    free(b->buf);
    free(b->ent);
    b->buf = NULL;
    b->ent = NULL;
}
#endif
