/* Test helper: drives filter 32025 through the real HDF5 library (C API), the way
 * /root/reference/examples/testCode.c:60-127 does, but loading the filter as a dynamic plugin
 * (HDF5_PLUGIN_PATH) instead of linking it.
 *   write    <file> <raw.bin> <rows> <cols> <chunk_rows> [cd...]   H5Dwrite through the filter (any cd_values)
 *   read     <file> <out.bin>                                      H5Dread through the filter
 *   chunks   <file> <prefix>                                       stored bytes of every chunk (H5Dread_chunk)
 *   writeraw <file> <rows> <cols> <chunk_rows> <M> <L> <prefix>    H5Dwrite_chunk of pre-encoded chunks
 * Environment H5RT_CHUNK_COLS=<n>: chunks of <chunk_rows> x <n> instead of full-width rows (write; chunks then dumps
 * the chunk grid row-major) -- /root/reference/examples/testCode.c:15-18 uses 32768 x 5 chunks of a 65536 x 10 dataset.
 */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FILTER 32025
#define CHECK(x) do { if ((x) < 0) { fprintf(stderr, "HDF5 call failed: %s (line %d)\n", #x, __LINE__); return 2; } } while (0)

static void *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    *n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc(*n ? *n : 1);
    if (fread(p, 1, *n, f) != *n) { fclose(f); free(p); return NULL; }
    fclose(f);
    return p;
}

static hid_t make_dcpl_n(hsize_t chunk_rows, hsize_t cols, size_t ncd, const unsigned *cd) {
    hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
    hsize_t chunk[2] = {chunk_rows, cols};
    if (getenv("H5RT_CHUNK_COLS")) chunk[1] = strtoull(getenv("H5RT_CHUNK_COLS"), 0, 10);
    H5Pset_chunk(dcpl, 2, chunk);
    if (H5Pset_filter(dcpl, FILTER, H5Z_FLAG_MANDATORY, ncd, cd) < 0) return -1;
    return dcpl;
}

static hid_t make_dcpl(hsize_t chunk_rows, hsize_t cols, unsigned M, unsigned L) {
    const unsigned cd[2] = {M, L};
    return make_dcpl_n(chunk_rows, cols, 2, cd);
}

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    if (!strcmp(argv[1], "write") && argc >= 7) {
        hsize_t rows = strtoull(argv[4], 0, 10), cols = strtoull(argv[5], 0, 10), crows = strtoull(argv[6], 0, 10);
        size_t n;
        short *raw = slurp(argv[3], &n);
        if (!raw || n != rows * cols * 2) { fprintf(stderr, "bad raw file\n"); return 1; }
        hid_t file = H5Fcreate(argv[2], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        hsize_t dims[2] = {rows, cols};
        hid_t space = H5Screate_simple(2, dims, NULL);
        unsigned cd[80];
        size_t ncd = 0;
        for (int i = 7; i < argc && ncd < 80; ++i) cd[ncd++] = (unsigned)strtoul(argv[i], 0, 10);
        hid_t dcpl = make_dcpl_n(crows, cols, ncd, cd);
        CHECK(dcpl);
        hid_t dset = H5Dcreate(file, "test", H5T_NATIVE_SHORT, space, H5P_DEFAULT, dcpl, H5P_DEFAULT);
        CHECK(dset);
        CHECK(H5Dwrite(dset, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw));
        H5Pclose(dcpl); H5Dclose(dset); H5Sclose(space); CHECK(H5Fclose(file));
        return 0;
    }
    if (!strcmp(argv[1], "read") && argc == 4) {
        hid_t file = H5Fopen(argv[2], H5F_ACC_RDONLY, H5P_DEFAULT);
        CHECK(file);
        hid_t dset = H5Dopen(file, "test", H5P_DEFAULT);
        CHECK(dset);
        hid_t space = H5Dget_space(dset);
        hsize_t dims[2];
        H5Sget_simple_extent_dims(space, dims, NULL);
        short *buf = malloc(dims[0] * dims[1] * 2);
        CHECK(H5Dread(dset, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf));
        FILE *f = fopen(argv[3], "wb");
        fwrite(buf, 2, dims[0] * dims[1], f);
        fclose(f);
        H5Sclose(space); H5Dclose(dset); H5Fclose(file);
        return 0;
    }
    if (!strcmp(argv[1], "chunks") && argc == 4) {
        hid_t file = H5Fopen(argv[2], H5F_ACC_RDONLY, H5P_DEFAULT);
        CHECK(file);
        hid_t dset = H5Dopen(file, "test", H5P_DEFAULT);
        CHECK(dset);
        hid_t space = H5Dget_space(dset), dcpl = H5Dget_create_plist(dset);
        hsize_t dims[2], chunk[2];
        H5Sget_simple_extent_dims(space, dims, NULL);
        H5Pget_chunk(dcpl, 2, chunk);
        int idx = 0;
        for (hsize_t r = 0; r < dims[0]; r += chunk[0])
        for (hsize_t c0 = 0; c0 < dims[1]; c0 += chunk[1], ++idx) {
            hsize_t off[2] = {r, c0}, nbytes = 0;
            uint32_t mask = 0;
            CHECK(H5Dget_chunk_storage_size(dset, off, &nbytes));
            void *buf = malloc(nbytes);
            CHECK(H5Dread_chunk(dset, H5P_DEFAULT, off, &mask, buf));
            char name[4096];
            snprintf(name, sizeof name, "%s.%d", argv[3], idx);
            FILE *f = fopen(name, "wb");
            fwrite(buf, 1, nbytes, f);
            fclose(f);
            free(buf);
        }
        printf("%d\n", idx);
        return 0;
    }
    if (!strcmp(argv[1], "writeraw") && argc == 9) {
        hsize_t rows = strtoull(argv[3], 0, 10), cols = strtoull(argv[4], 0, 10), crows = strtoull(argv[5], 0, 10);
        hid_t file = H5Fcreate(argv[2], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
        hsize_t dims[2] = {rows, cols};
        hid_t space = H5Screate_simple(2, dims, NULL);
        hid_t dcpl = make_dcpl(crows, cols, (unsigned)atoi(argv[6]), (unsigned)atoi(argv[7]));
        CHECK(dcpl);
        hid_t dset = H5Dcreate(file, "test", H5T_NATIVE_SHORT, space, H5P_DEFAULT, dcpl, H5P_DEFAULT);
        CHECK(dset);
        int idx = 0;
        for (hsize_t r = 0; r < rows; r += crows, ++idx) {
            char name[4096];
            snprintf(name, sizeof name, "%s.%d", argv[8], idx);
            size_t n;
            void *buf = slurp(name, &n);
            if (!buf) { fprintf(stderr, "missing %s\n", name); return 1; }
            hsize_t off[2] = {r, 0};
            CHECK(H5Dwrite_chunk(dset, H5P_DEFAULT, 0, off, n, buf));
            free(buf);
        }
        H5Pclose(dcpl); H5Dclose(dset); H5Sclose(space); CHECK(H5Fclose(file));
        return 0;
    }
    fprintf(stderr, "bad arguments\n");
    return 1;
}
