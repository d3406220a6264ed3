// bert-main — command-line smoke tool for libbert.so: tokenize one prompt, print the ids and pieces, evaluate
// it, print the embedding and the timings.  Same command line (bert_params_parse) and the same stdout shape as
// the reference's demo (examples/main.cpp:8-77), so scripts that scrape it keep working; the implementation
// only uses the public C API of include/bert.h and a steady clock.
#include <chrono>
#include <cstdio>
#include <vector>

#include "bert.h"

namespace {
double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
}  // namespace

int main(int argc, char **argv) {
    const auto t_main = std::chrono::steady_clock::now();

    bert_params params;
    params.model = "models/all-MiniLM-L6-v2/ggml-model-f32.bin";
    if (!bert_params_parse(argc, argv, params)) return 1;

    const auto t_load = std::chrono::steady_clock::now();
    bert_ctx *ctx = bert_load_from_file(params.model);
    if (ctx == nullptr) {
        fprintf(stderr, "main: failed to load model from '%s'\n", params.model);
        return 1;
    }
    const double load_ms = ms_since(t_load);

    const auto t_eval = std::chrono::steady_clock::now();
    const int32_t cap = bert_n_max_tokens(ctx);
    std::vector<bert_vocab_id> ids((size_t)cap);
    int32_t n_ids = 0;
    bert_tokenize(ctx, params.prompt, ids.data(), &n_ids, cap);
    ids.resize((size_t)n_ids);

    printf("main: number of tokens in prompt = %zu\n\n", ids.size());
    printf("[");
    for (bert_vocab_id id : ids) printf("%d, ", id);
    printf("]\n");
    for (bert_vocab_id id : ids) printf("%d -> %s\n", id, bert_vocab_id_to_token(ctx, id));

    std::vector<float> emb((size_t)bert_n_embd(ctx), 0.0f);
    bert_eval(ctx, params.n_threads, ids.data(), n_ids, emb.data());
    const double eval_ms = ms_since(t_eval);

    printf("[");
    for (float e : emb) printf("%1.4f, ", e);
    printf("]\n");

    printf("\n\n");
    printf("main:     load time = %8.2f ms\n", load_ms);
    printf("main:  eval time = %8.2f ms / %.2f ms per token\n", eval_ms, eval_ms / (ids.empty() ? 1 : ids.size()));
    printf("main:    total time = %8.2f ms\n", ms_since(t_main));

    bert_free(ctx);
    return 0;
}
