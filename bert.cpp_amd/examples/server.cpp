// bert-server — TCP embedding server with the wire protocol of the reference (examples/server.cpp:36-124,
// client side examples/sample_client.py:9-22):
//   on connect the server sends int32 n_embd; then every read() of <= 32768 bytes from a client is one
//   request text (no framing) answered by n_embd little-endian f32; an empty read closes the connection.
// What differs is the serving loop.  The reference accepts ONE client and encodes one text per round trip;
// a GPU engine wants batches, so this server poll()s any number of clients, takes every request that is
// readable in the same poll round and evaluates them as ONE bert_encode_batch call (results are identical
// to per-request bert_encode — that is bert_encode_batch's contract), then answers each client.  A single
// client sees exactly the reference's behaviour.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

#include <cerrno>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "bert.h"
#include "bert_hip.h"

namespace {

constexpr size_t kMaxRequest = 1 << 15;   // same request size limit as the reference's receive buffer

bool send_all(int fd, const void *data, size_t n) {
    const char *p = (const char *)data;
    while (n > 0) {
        ssize_t w = send(fd, p, n, MSG_NOSIGNAL);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += w;
        n -= (size_t)w;
    }
    return true;
}

volatile sig_atomic_t g_stop = 0;
void on_signal(int) { g_stop = 1; }

}  // namespace

int main(int argc, char **argv) {
    bert_params params;
    params.model = "models/all-MiniLM-L6-v2/ggml-model-q4_0.bin";
    if (!bert_params_parse(argc, argv, params)) return 1;

    bert_ctx *ctx = bert_load_from_file(params.model);
    if (ctx == nullptr) {
        fprintf(stderr, "main: failed to load model from '%s'\n", params.model);
        return 1;
    }
    const int32_t n_embd = bert_n_embd(ctx);

    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_signal;
    sigaction(SIGINT, &sa, nullptr);
    sigaction(SIGTERM, &sa, nullptr);

    int listener = socket(AF_INET, SOCK_STREAM, 0);
    if (listener < 0) { perror("socket"); return 1; }
    int one = 1;
    setsockopt(listener, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in addr;
    memset(&addr, 0, sizeof(addr));
    addr.sin_family = AF_INET;
    addr.sin_addr.s_addr = htonl(INADDR_ANY);
    addr.sin_port = htons((uint16_t)params.port);
    if (bind(listener, (sockaddr *)&addr, sizeof(addr)) < 0) { perror("bind"); return 1; }
    if (listen(listener, 64) < 0) { perror("listen"); return 1; }
    socklen_t alen = sizeof(addr);
    getsockname(listener, (sockaddr *)&addr, &alen);   // --port 0 picks a free port; report the real one
    printf("Server running on port %d with %d threads\n", (int)ntohs(addr.sin_port), params.n_threads);
    fflush(stdout);

    std::vector<pollfd> fds{{listener, POLLIN, 0}};
    std::vector<std::string> texts;          // requests of this poll round
    std::vector<int> owners;                 // fds index of each request
    std::vector<float> out;
    std::vector<const char *> text_ptrs;
    std::vector<float *> out_ptrs;
    std::vector<char> buf(kMaxRequest);

    while (!g_stop) {
        int ready = poll(fds.data(), (nfds_t)fds.size(), 500);
        if (ready < 0) {
            if (errno == EINTR) continue;
            perror("poll");
            break;
        }
        if (ready == 0) continue;

        texts.clear();
        owners.clear();
        std::vector<int> closing;
        for (size_t i = 1; i < fds.size(); ++i) {
            if (!(fds[i].revents & (POLLIN | POLLHUP | POLLERR))) continue;
            ssize_t n = read(fds[i].fd, buf.data(), buf.size());
            if (n <= 0) {
                closing.push_back((int)i);
                continue;
            }
            texts.emplace_back(buf.data(), (size_t)n);
            owners.push_back((int)i);
        }

        if (!texts.empty()) {
            const int32_t n = (int32_t)texts.size();
            out.assign((size_t)n * n_embd, 0.0f);
            text_ptrs.resize(n);
            out_ptrs.resize(n);
            for (int32_t r = 0; r < n; ++r) {
                text_ptrs[r] = texts[r].c_str();
                out_ptrs[r] = out.data() + (size_t)r * n_embd;
            }
            // bert_hip_encode_batch = bert_encode_batch with a result: the requests it could not evaluate (device error) get
            // no reply at all — their connections are closed instead of being sent the zero-filled rows
            const int32_t done = bert_hip_encode_batch(ctx, params.n_threads, n, text_ptrs.data(), out_ptrs.data());
            for (int32_t r = 0; r < n; ++r)
                if (r >= done || !send_all(fds[owners[r]].fd, out_ptrs[r], sizeof(float) * (size_t)n_embd)) closing.push_back(owners[r]);
        }

        if (fds[0].revents & POLLIN) {
            int c = accept(listener, nullptr, nullptr);
            if (c >= 0) {
                setsockopt(c, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
                // one client that stops reading must not stall the others: a send that blocks for 2 s fails, the client is dropped
                timeval tv{2, 0};
                setsockopt(c, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
                if (send_all(c, &n_embd, sizeof(n_embd))) {
                    fds.push_back({c, POLLIN, 0});
                    printf("New connection\n");
                    fflush(stdout);
                } else {
                    close(c);
                }
            }
        }

        // drop closed clients, highest index first so the remaining indices stay valid
        for (size_t a = 0; a < closing.size(); ++a)
            for (size_t b = a + 1; b < closing.size(); ++b)
                if (closing[b] > closing[a]) std::swap(closing[a], closing[b]);
        int last = -1;
        for (int idx : closing) {
            if (idx == last) continue;
            last = idx;
            close(fds[idx].fd);
            fds.erase(fds.begin() + idx);
        }
    }

    for (size_t i = 1; i < fds.size(); ++i) close(fds[i].fd);
    close(listener);
    bert_free(ctx);
    return 0;
}
