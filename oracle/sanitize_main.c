/* Sanitiser driver of the CPU oracle (test infrastructure): runs seeded rollouts of hsr_oracle.c under AddressSanitizer /
 * UndefinedBehaviorSanitizer (sanitised run of the CPU restatement; GPU sanitizers are not available).
 * usage: oracle_asan <model.hsrm> <inputs.bin> <nsteps>
 * inputs.bin: int32 n, nq, nu, then n x nq doubles (qpos0) and n x nu doubles (ctrl).  Prints a checksum of the final states. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct ho_model ho_model;
typedef struct ho_data ho_data;
ho_model *ho_model_load(const void *blob, size_t len);
void ho_model_free(ho_model *m);
ho_data *ho_data_new(const ho_model *m);
void ho_data_free(ho_data *d);
void ho_reset(const ho_model *m, ho_data *d);
void ho_step(const ho_model *m, ho_data *d);
double *ho_qpos(ho_data *d);
double *ho_qvel(ho_data *d);
double *ho_ctrl(ho_data *d);
int ho_bad(const ho_data *d);
int ho_model_size(const ho_model *m, int which);

static void *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    void *p = malloc((size_t)n);
    if (fread(p, 1, (size_t)n, f) != (size_t)n) { perror("read"); exit(2); }
    fclose(f); *len = (size_t)n;
    return p;
}

int main(int argc, char **argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s model.hsrm inputs.bin nsteps\n", argv[0]); return 2; }
    size_t blen, ilen;
    void *blob = slurp(argv[1], &blen);
    char *in = slurp(argv[2], &ilen);
    const int nsteps = atoi(argv[3]);
    int hdr[3];
    memcpy(hdr, in, sizeof hdr);
    const int n = hdr[0], nq = hdr[1], nu = hdr[2];
    const double *q0 = (const double *)(in + sizeof hdr), *ctrl = q0 + (size_t)n * nq;
    ho_model *m = ho_model_load(blob, blen);
    if (!m) { fprintf(stderr, "model load failed\n"); return 3; }
    const int nv = ho_model_size(m, 1);
    double sum = 0;
    for (int e = 0; e < n; e++) {
        ho_data *d = ho_data_new(m);
        ho_reset(m, d);
        memcpy(ho_qpos(d), q0 + (size_t)e * nq, sizeof(double) * (size_t)nq);
        memcpy(ho_ctrl(d), ctrl + (size_t)e * nu, sizeof(double) * (size_t)nu);
        for (int k = 0; k < nsteps; k++) ho_step(m, d);
        for (int i = 0; i < nq; i++) sum += ho_qpos(d)[i];
        for (int i = 0; i < nv; i++) sum += 1e-3 * ho_qvel(d)[i];
        if (ho_bad(d)) { fprintf(stderr, "env %d went bad\n", e); return 4; }
        ho_data_free(d);
    }
    ho_model_free(m);
    free(blob); free(in);
    printf("%.12e\n", sum);
    return 0;
}
