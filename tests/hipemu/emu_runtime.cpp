// tests/hipemu/emu_runtime.cpp -- the launch loop of the host stand-in (see hip/hip_runtime.h); test infrastructure only.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <mutex>

namespace hipemu {

int g_ncu = 2; // "CUs" reported to persistent kernels: two workgroups, run one after the other
thread_local Ctx tl;

namespace {

// Worker threads live across launches (creating 1024 threads per workgroup dominated the run time): generation counter
// + condition variable; workers beyond the workgroup's size sit a generation out.
struct Pool {
    std::mutex m;
    std::condition_variable cv_start, cv_done;
    std::vector<std::thread> th;
    unsigned long gen = 0;
    int nactive = 0, remaining = 0;
    const std::function<void(int)>* job = nullptr;

    void worker(int id) {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)>* j = nullptr;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_start.wait(lk, [&] { return gen != seen; });
                seen = gen;
                if (id < nactive) {
                    j = job;
                }
            }
            if (j != nullptr) {
                (*j)(id);
                std::unique_lock<std::mutex> lk(m);
                if (--remaining == 0) {
                    cv_done.notify_all();
                }
            }
        }
    }
    void run(int n, const std::function<void(int)>& f) {
        {
            std::unique_lock<std::mutex> lk(m);
            while ((int)th.size() < n) {
                const int id = (int)th.size();
                th.emplace_back([this, id] { worker(id); });
                th.back().detach();
            }
            job = &f;
            nactive = n;
            remaining = n;
            gen++;
        }
        cv_start.notify_all();
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return remaining == 0; });
    }
};

Pool& pool() {
    static Pool* p = new Pool(); // (never destroyed: detached workers may outlive static destruction)
    return *p;
}

} // namespace

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    static std::mutex launch_mu; // one launch at a time (host threads of a concurrent test take turns)
    std::lock_guard<std::mutex> launch_lock(launch_mu);
    const int nthreads = (int)(block.x * block.y * block.z);
    const int nwaves = (nthreads + 63) / 64;
    std::vector<unsigned char> smem(smem_bytes + 64);
    for (unsigned bz = 0; bz < grid.z; bz++) {
        for (unsigned by = 0; by < grid.y; by++) {
            for (unsigned bx = 0; bx < grid.x; bx++) {
                Group g(nthreads);
                std::memset(smem.data(), 0xcd, smem.size()); // (LDS starts undefined on the hardware)
                g.smem = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(smem.data()) + 63) & ~(uintptr_t)63);
                for (int w = 0; w < nwaves; w++) {
                    g.waves.emplace_back(new Wave(std::min(64, nthreads - w * 64)));
                }
                const std::function<void(int)> job = [&](int t) {
                    tl = Ctx{};
                    tl.tid = dim3((unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y));
                    tl.bid = dim3(bx, by, bz);
                    tl.bdim = block;
                    tl.gdim = grid;
                    tl.g = &g;
                    tl.w = g.waves[t / 64].get();
                    tl.lane = t % 64;
                    body();
                    tl.w->bar.arrive_and_drop();
                    g.bar.arrive_and_drop();
                };
                pool().run(nthreads, job);
            }
        }
    }
}

} // namespace hipemu
