/* Example (plain C, the boundary exactly as a foreign-function binding sees it): Policy::eval (study-connect4/src/policies.rs:47-59)
 * from worker threads — every worker owns an evaluation context, the reference's one policy object per worker thread
 * (alpha_zero.rs:192-198) — on one engine whose weights were loaded once.
 *
 *   gcc -O2 -pthread -I include examples/policy_eval_worker.c -o policy_eval_worker -L synthesis_amd -lsynthesis_amd \
 *       -Wl,-rpath,$PWD/synthesis_amd
 *   ./policy_eval_worker weights.f32        (30,492 little-endian f32: l_1.weight, l_1.bias, ..., l_5.bias)
 *
 * Prints, per worker, the logits and the value of the position after the opening moves d, d+1, ... (column indices), evaluated
 * one position per call (what a Rust `impl Policy` does) and all at once with submit ... wait around other work. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "synthesis_amd.h"

#define WORKERS 3
#define POSITIONS 5

typedef struct {
    syn_engine* engine;
    int id;
    int rc;
    float logits[POSITIONS][9], value[POSITIONS][3], batch_logits[POSITIONS][9], batch_value[POSITIONS][3];
} worker_t;

/* connect4.rs:3-13: bit row + 7 * column; `my` is the side to move */
static void play(const int* cols, int n, uint64_t* my, uint64_t* op) {
    uint64_t a = 0, b = 0;   /* a = side to move */
    for (int i = 0; i < n; i++) {
        int h = 0;
        for (uint64_t c = ((a | b) >> (7 * cols[i])) & 0x7F; c; c &= c - 1) h++;
        const uint64_t mine = a | (1ull << (h + 7 * cols[i]));
        a = b;
        b = mine;
    }
    *my = a;
    *op = b;
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static void* work(void* arg) {
    worker_t* w = (worker_t*)arg;
    syn_eval_ctx* ctx = NULL;
    w->rc = syn_eval_ctx_create(w->engine, &ctx);
    if (w->rc != SYN_OK) return NULL;
    uint64_t my[POSITIONS], op[POSITIONS];
    int cols[POSITIONS];
    for (int i = 0; i < POSITIONS; i++) {
        cols[i] = (w->id + i) % 9;
        play(cols, i + 1, &my[i], &op[i]);
    }
    for (int i = 0; i < POSITIONS && w->rc == SYN_OK; i++)   /* Policy::eval, one state per call */
        w->rc = syn_eval_ctx_eval(ctx, &my[i], &op[i], 1, w->logits[i], w->value[i]);
    if (w->rc == SYN_OK) w->rc = syn_eval_ctx_submit(ctx, my, op, POSITIONS);   /* returns at once: the batch is on its way */
    /* ... a self-play worker advances its other trees here ... */
    if (w->rc == SYN_OK) w->rc = syn_eval_ctx_wait(ctx, &w->batch_logits[0][0], &w->batch_value[0][0]);
    if (w->rc != SYN_OK) fprintf(stderr, "worker %d: %s\n", w->id, syn_eval_ctx_last_error(ctx));
    syn_eval_ctx_destroy(ctx);
    return NULL;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s weights.f32\n", argv[0]); return 2; }
    static float blob[30492];
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(blob, 4, 30492, f) != 30492) { fprintf(stderr, "cannot read 30492 floats from %s\n", argv[1]); return 2; }
    fclose(f);
    syn_engine_config cfg = {0};
    cfg.concurrent_games = 256;
    cfg.max_explores = 64;
    syn_engine* engine = NULL;
    if (syn_engine_create(&cfg, 0, &engine) != SYN_OK) { fprintf(stderr, "%s\n", syn_last_error(NULL)); return 1; }
    if (syn_load_weights(engine, blob, 30492) != SYN_OK) { fprintf(stderr, "%s\n", syn_last_error(engine)); return 1; }
    worker_t workers[WORKERS];
    pthread_t threads[WORKERS];
    for (int i = 0; i < WORKERS; i++) {
        workers[i].engine = engine;
        workers[i].id = i;
        pthread_create(&threads[i], NULL, work, &workers[i]);
    }
    int rc = 0;
    for (int i = 0; i < WORKERS; i++) {
        pthread_join(threads[i], NULL);
        if (workers[i].rc != SYN_OK) rc = 1;
    }
    for (int i = 0; i < WORKERS && rc == 0; i++)
        for (int p = 0; p < POSITIONS; p++) {
            printf("worker %d position %d:", i, p);
            for (int k = 0; k < 9; k++) printf(" %08x", bits(workers[i].logits[p][k]));
            for (int k = 0; k < 3; k++) printf(" %08x", bits(workers[i].value[p][k]));
            int same = 1;
            for (int k = 0; k < 9; k++) same &= workers[i].logits[p][k] == workers[i].batch_logits[p][k];
            for (int k = 0; k < 3; k++) same &= workers[i].value[p][k] == workers[i].batch_value[p][k];
            printf(" batch %s\n", same ? "same" : "DIFFERENT");
        }
    syn_engine_destroy(engine);
    return rc;
}
