// Batched solves of independent LPs (BASELINE config 4: the Netlib suite, one LP per GPU in flight).
//
// The reference is single-threaded and solves one LP per call (tests/netlib/mod.rs:47-71); independent LPs are the only
// thing that shards (each pivot of one LP is a serial dependency chain).  A batch owns `workers_per_device` host threads per
// device, each with one resident handle (= one HIP stream) per LP of the batch; a run serves a ticket queue: ticket t is LP
// schedule[t], tickets are drawn by an atomic fetch-add -- or by the caller's `next_ticket`, so that several processes (one
// rank per GPU under torch.distributed) can share ONE queue.  No collective and no data exchange between workers.
// Built on the public C ABI only (relp_create / relp_load_model / relp_solve_relaxation): what a caller could write itself.
#include "solver.hpp"
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/relp_amd.h"

namespace {
double now_seconds() {
    using clock = std::chrono::steady_clock;
    return std::chrono::duration<double>(clock::now().time_since_epoch()).count();
}
void set_error(char* error, int32_t capacity, const std::string& text) {
    if (!error || capacity <= 0) return;
    const size_t n = std::min<size_t>(text.size(), (size_t)capacity - 1);
    std::memcpy(error, text.data(), n);
    error[n] = 0;
}
}  // namespace

struct relp_batch {
    int n_models = 0;
    std::vector<int32_t> worker_device;
    std::vector<std::vector<relp_handle*>> handles;  // [worker][model]
    std::vector<std::string> exact;                  // per ticket of the last run
    std::string error;
};

extern "C" {

int32_t relp_batch_create(const relp_model* const* models, int32_t n_models, const relp_options* options, const int32_t* devices,
                          int32_t n_devices, int32_t workers_per_device, relp_batch** out, char* error, int32_t error_capacity) {
    if (!models || n_models <= 0 || !options || !devices || n_devices <= 0 || workers_per_device <= 0 || !out) return RELP_ERR_ARGUMENT;
    relp_options adopted;  // (as many bytes as the caller's header had: relp_options.struct_size)
    if (relp::adopt_options(options, &adopted) != RELP_OK) return RELP_ERR_ARGUMENT;
    auto batch = std::make_unique<relp_batch>();
    batch->n_models = n_models;
    for (int d = 0; d < n_devices; ++d)
        for (int w = 0; w < workers_per_device; ++w) batch->worker_device.push_back(devices[d]);
    const int n_workers = (int)batch->worker_device.size();
    batch->handles.assign(n_workers, std::vector<relp_handle*>(n_models, nullptr));
    // every worker uploads its own copies (one thread per worker: the uploads of different streams overlap)
    std::vector<int32_t> status(n_workers, RELP_OK);
    std::vector<std::string> messages(n_workers);
    std::vector<std::thread> threads;
    for (int w = 0; w < n_workers; ++w)
        threads.emplace_back([&, w] {
            relp_options o = adopted;
            o.device = batch->worker_device[w];
            for (int k = 0; k < n_models && status[w] == RELP_OK; ++k) {
                relp_handle* h = nullptr;
                int32_t s = relp_create(&o, &h);
                if (s == RELP_OK) {
                    batch->handles[w][k] = h;
                    s = relp_load_model(h, models[k]);
                    if (s != RELP_OK) messages[w] = std::string("model ") + std::to_string(k) + ": " + relp_last_error(h);
                } else {
                    messages[w] = "relp_create failed (no usable HIP device? the product has no CPU fallback)";
                }
                status[w] = s;
            }
        });
    for (auto& t : threads) t.join();
    for (int w = 0; w < n_workers; ++w)
        if (status[w] != RELP_OK) {
            set_error(error, error_capacity, messages[w]);
            const int32_t s = status[w];
            relp_batch* raw = batch.release();
            relp_batch_destroy(raw);
            return s;
        }
    *out = batch.release();
    return RELP_OK;
}

int32_t relp_batch_destroy(relp_batch* batch) {
    if (!batch) return RELP_ERR_ARGUMENT;
    for (auto& per_worker : batch->handles)
        for (relp_handle* h : per_worker)
            if (h) relp_destroy(h);
    delete batch;
    return RELP_OK;
}

int32_t relp_batch_workers(const relp_batch* batch, int32_t* n_workers) {
    if (!batch || !n_workers) return RELP_ERR_ARGUMENT;
    *n_workers = (int32_t)batch->worker_device.size();
    return RELP_OK;
}

int32_t relp_batch_run(relp_batch* batch, const int32_t* schedule, int64_t n_tickets, int64_t (*next_ticket)(void* user), void* user,
                       relp_batch_entry* entries, relp_batch_worker* workers, double* makespan_seconds) {
    if (!batch || !schedule || n_tickets <= 0 || !entries) return RELP_ERR_ARGUMENT;
    for (int64_t t = 0; t < n_tickets; ++t)
        if (schedule[t] < 0 || schedule[t] >= batch->n_models) return RELP_ERR_ARGUMENT;
    const int n_workers = (int)batch->worker_device.size();
    for (int64_t t = 0; t < n_tickets; ++t) {
        entries[t] = relp_batch_entry{};
        entries[t].status = -1;  // not served by this batch (another process drew the ticket)
        entries[t].worker = entries[t].device = -1;
        entries[t].model = schedule[t];
    }
    batch->exact.assign((size_t)n_tickets, std::string());
    std::atomic<int64_t> head{0};
    // a ticket is served once: a `next_ticket` that repeats one (or a binding whose callback failed and returned 0 every time) ends
    // the worker that drew the repeat instead of letting two threads write entries[t] and the same handle's state concurrently
    std::vector<std::atomic<char>> claimed((size_t)n_tickets);
    for (auto& c : claimed) c.store(0, std::memory_order_relaxed);
    std::atomic<int> repeats{0};
    std::vector<relp_batch_worker> stats(n_workers);
    const double t0 = now_seconds();
    std::vector<std::thread> threads;
    for (int w = 0; w < n_workers; ++w)
        threads.emplace_back([&, w] {
            relp_batch_worker& me = stats[w];
            me = relp_batch_worker{};
            me.device = batch->worker_device[w];
            for (;;) {
                const double t_ask = now_seconds();
                const int64_t t = next_ticket ? next_ticket(user) : head.fetch_add(1, std::memory_order_relaxed);
                me.queue_seconds += now_seconds() - t_ask;
                if (t < 0 || t >= n_tickets) break;
                if (claimed[(size_t)t].exchange(1, std::memory_order_acq_rel)) {
                    repeats.fetch_add(1, std::memory_order_relaxed);
                    break;
                }
                relp_handle* h = batch->handles[w][schedule[t]];
                relp_batch_entry& e = entries[t];
                e.worker = w;
                e.device = me.device;
                e.start_seconds = now_seconds() - t0;
                e.status = relp_solve_relaxation(h, &e.result);
                e.end_seconds = now_seconds() - t0;
                if (e.status == RELP_OK && e.result.certified) {
                    int32_t length = 0;
                    if (relp_get_objective_exact(h, nullptr, 0, &length) == RELP_OK && length > 0) {
                        std::string text((size_t)length + 1, '\0');
                        relp_get_objective_exact(h, &text[0], length + 1, &length);
                        text.resize((size_t)length);
                        batch->exact[(size_t)t] = text;
                    }
                }
                me.tickets += 1;
                me.pivots += e.result.pivots_phase_one + e.result.pivots_phase_two;
                me.busy_seconds += e.end_seconds - e.start_seconds;
            }
            me.finish_seconds = now_seconds() - t0;
        });
    for (auto& t : threads) t.join();
    const double makespan = now_seconds() - t0;
    if (makespan_seconds) *makespan_seconds = makespan;
    if (workers)
        for (int w = 0; w < n_workers; ++w) {
            workers[w] = stats[w];
            workers[w].idle_seconds = makespan - stats[w].busy_seconds;
        }
    return repeats.load() ? RELP_ERR_ARGUMENT : RELP_OK;  // the entries served so far are valid; the source handed a ticket out twice
}

int32_t relp_batch_get_objective_exact(const relp_batch* batch, int64_t ticket, char* buffer, int32_t capacity, int32_t* length) {
    if (!batch || ticket < 0 || ticket >= (int64_t)batch->exact.size()) return RELP_ERR_ARGUMENT;
    const std::string& s = batch->exact[(size_t)ticket];
    if (length) *length = (int32_t)s.size();
    if (s.empty()) return RELP_ERR_STATE;
    if (buffer && capacity > 0) {
        const int32_t nbytes = std::min<int32_t>((int32_t)s.size(), capacity - 1);
        std::memcpy(buffer, s.data(), nbytes);
        buffer[nbytes] = 0;
    }
    return RELP_OK;
}

int32_t relp_batch_handle(const relp_batch* batch, int32_t worker, int32_t model, relp_handle** out) {
    if (!batch || !out || worker < 0 || worker >= (int32_t)batch->handles.size() || model < 0 || model >= batch->n_models) return RELP_ERR_ARGUMENT;
    *out = batch->handles[worker][model];
    return RELP_OK;
}

}  // extern "C"
