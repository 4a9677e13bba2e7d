// Host-side profile stages (cut-offs, propagation, abundance writer) on a synthetic 5000-reference case:
// timing loop and the driver of the ASan/UBSan run in scripts/sanitize_host.sh.
#include "../../slimm_amd/csrc/host_profile.hpp"
#include <chrono>
#include <cstdio>
#include <random>
using namespace slimm;
int main(){
  const uint32_t R=5000; HostConfig hc; hc.n_refs=R; hc.bin_width=1000; hc.avg_read_len=100;
  std::mt19937 g(1);
  for(uint32_t i=0;i<R;i++){ hc.ref_len.push_back(2000000+g()%4000000); uint32_t l[8]={10000000+i,1000000+i/2,500000+i/10,200000+i/50,100000+i/200,50000+i/1000,10000+i/2000,2}; for(int k=0;k<8;k++){hc.lineage.push_back(l[k]);} }
  for(uint32_t i=0;i<R;i++){ uint32_t l[8]={10000000+i,1000000+i/2,500000+i/10,200000+i/50,100000+i/200,50000+i/1000,10000+i/2000,2}; for(int k=0;k<8;k++){ hc.tax_id.push_back(l[k]); hc.tax_rank.push_back(k); hc.tax_name.push_back("name_"+std::to_string(l[k])); } }
  HostProfile h(hc);
  std::vector<uint32_t> rc(R,0),uc(R,0),nz(R,0),nzu(R,0),u2(R,0),marks(R,0); std::vector<uint32_t> lca(h.n_taxa_dense(),0);
  for(uint32_t i=0;i<R;i+=25){ rc[i]=1000+g()%100000; uc[i]=rc[i]/2; nz[i]=h.nbins()[i]/2; nzu[i]=nz[i]/2; u2[i]=uc[i]; rc[i+1]=50; uc[i+1]=5; nz[i+1]=20; nzu[i+1]=3; marks[i]=2; marks[i+1]=2; lca[h.lineage_dense()[i*8+1]]=100; }
  auto T0=std::chrono::steady_clock::now(); double tv=0,tp=0,tw=0; int N=500;
  for(int it=0;it<N;it++){
    h.reset(); h.reset_cutoffs();
    h.set_coverage(rc.data(),uc.data(),nz.data(),nzu.data(),5000000,2000000);
    auto a=std::chrono::steady_clock::now();
    h.compute_valid();
    auto b=std::chrono::steady_clock::now();
    h.set_partials(u2.data(),lca.data(),marks.data(),nullptr,0);
    h.propagate();
    auto c=std::chrono::steady_clock::now();
    const std::string& s=h.write_abundance();
    auto d=std::chrono::steady_clock::now();
    if(it==0) printf("%zu bytes\n", s.size());
    tv+=std::chrono::duration<double,std::micro>(b-a).count(); tp+=std::chrono::duration<double,std::micro>(c-b).count(); tw+=std::chrono::duration<double,std::micro>(d-c).count();
  }
  printf("valid %.1f us, propagate %.1f us, write %.1f us\n", tv/N,tp/N,tw/N);
}
