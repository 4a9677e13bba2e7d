// How many cores does the box really grant?  Build and run: g++ -O2 -pthread -o scripts/ubench/spin scripts/ubench/spin.cpp && scripts/ubench/spin
#include <thread>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdint>
int main(){ for (int nt : {1,8,32,64,128,256}) { auto t0=std::chrono::steady_clock::now(); std::vector<std::thread> p; std::vector<uint64_t> out(nt);
 for (int t=0;t<nt;++t) p.emplace_back([&,t]{ uint64_t x=t+1; for (uint64_t i=0;i<400000000ull;++i) x = x*6364136223846793005ull+1442695040888963407ull; out[t]=x;});
 for (auto& th: p) th.join(); printf("%d threads %.3f s\n", nt, std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count()); } }
