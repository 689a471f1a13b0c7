// tests/hipemu/emu_runtime.cpp -- the launch loop of the host stand-in (see hip/hip_runtime.h); test infrastructure only.
#include <hip/hip_runtime.h>

namespace hipemu {

int g_ncu = 2; // "CUs" reported to persistent kernels: two workgroups, run one after the other
thread_local Ctx tl;

void launch(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body) {
    const int nthreads = (int)(block.x * block.y * block.z);
    const int nwaves = (nthreads + 63) / 64;
    for (unsigned bz = 0; bz < grid.z; bz++) {
        for (unsigned by = 0; by < grid.y; by++) {
            for (unsigned bx = 0; bx < grid.x; bx++) {
                Group g(nthreads);
                std::vector<unsigned char> smem(smem_bytes + 64, 0xcd); // (LDS starts undefined on the hardware)
                g.smem = reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(smem.data()) + 63) & ~(uintptr_t)63);
                for (int w = 0; w < nwaves; w++) {
                    g.waves.emplace_back(new Wave(std::min(64, nthreads - w * 64)));
                }
                std::vector<std::thread> th;
                th.reserve(nthreads);
                for (int t = 0; t < nthreads; t++) {
                    th.emplace_back([&, t]() {
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
                    });
                }
                for (auto& x : th) {
                    x.join();
                }
            }
        }
    }
}

} // namespace hipemu
