// Butterfly-throughput probe: variants of the exact Harvey/Shoup butterfly with per-lane twiddles from memory.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_bfly.hip -o tools/ubench_bfly
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64; typedef unsigned int u32;
struct MulOp { u64 op, quo; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ u64 mulA(u64 x, MulOp w, u64 q) { return w.op * x - __umul64hi(x, w.quo) * q; }
__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) { u64 r, cy; asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(cy) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ u64 mad64z(u32 a, u32 b) { u64 r, cy; asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(cy) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ u64 shr32(u64 a) { u64 r; asm("v_lshrrev_b64 %0, 32, %1" : "=v"(r) : "v"(a)); return r; }
__device__ __forceinline__ u64 add64(u64 a, u64 b) { u64 r; asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// exact hi64(x*a) as halves
__device__ __forceinline__ void mulhi_halves(u32 x0, u32 x1, u32 a0, u32 a1, u32 &h0, u32 &h1) {
  u64 T = shr32(mad64z(x0,a0));
  u64 A = mad64(x0,a1,T);
  u64 B = mad64z(x1,a0);
  u64 H = mad64z(x1,a1);
  u32 j, s, c;
  asm("v_add_co_u32 %0, vcc, %5, %6\n\tv_addc_co_u32 %1, vcc, %7, %8, vcc\n\tv_addc_co_u32 %2, vcc, 0, 0, vcc\n\t"
      "v_add_co_u32 %3, vcc, %9, %1\n\tv_addc_co_u32 %4, vcc, %10, %2, vcc"
      : "=&v"(j), "=&v"(s), "=&v"(c), "=&v"(h0), "=v"(h1)
      : "v"((u32)B), "v"((u32)A), "v"((u32)(B>>32)), "v"((u32)(A>>32)), "v"((u32)H), "v"((u32)(H>>32)) : "vcc");
}
__device__ __forceinline__ u64 mulD(u64 x, MulOp w, u64 nq) {
  u32 x0=(u32)x, x1=(u32)(x>>32), w0=(u32)w.op, w1=(u32)(w.op>>32), n0=(u32)nq, n1=(u32)(nq>>32), h0, h1;
  mulhi_halves(x0,x1,(u32)w.quo,(u32)(w.quo>>32),h0,h1);
  u64 E = mad64(h0,n0, mad64z(x0,w0));
  u64 M = mad64(h1,n0, mad64(h0,n1, mad64(x1,w0, mad64z(x0,w1))));
  u32 hi = (u32)(E>>32) + (u32)M;
  return ((u64)hi<<32) | (u32)E;
}
// C++ only: exact hi64 as halves with carry builtins
__device__ __forceinline__ u64 mulE(u64 x, MulOp w, u64 nq) {
  u32 x0=(u32)x, x1=(u32)(x>>32), w0=(u32)w.op, w1=(u32)(w.op>>32), a0=(u32)w.quo, a1=(u32)(w.quo>>32), n0=(u32)nq, n1=(u32)(nq>>32);
  u64 A = (u64)x0*a1 + __umulhi(x0,a0);
  u64 B = (u64)x1*a0;
  u64 H = (u64)x1*a1;
  u32 c1, c2, c3, c4;
  (void)__builtin_addc((u32)B, (u32)A, 0u, &c1);
  u32 s = __builtin_addc((u32)(B>>32), (u32)(A>>32), c1, &c2);
  u32 h0 = __builtin_addc((u32)H, s, 0u, &c3);
  u32 h1 = __builtin_addc((u32)(H>>32), c2, c3, &c4);
  u64 E = (u64)x0*w0 + (u64)h0*n0;
  u32 m = x0*w1 + x1*w0 + h0*n1 + h1*n0;
  u32 hi = (u32)(E>>32) + m;
  return ((u64)hi<<32) | (u32)E;
}
// variant F: compiler mulhi, mad chains for the low product (empty-asm barrier keeps the 64-bit chain)
__device__ __forceinline__ u64 mulF(u64 x, MulOp w, u64 nq) {
  u64 h = __umul64hi(x, w.quo);
  u32 x0=(u32)x, x1=(u32)(x>>32), w0=(u32)w.op, w1=(u32)(w.op>>32), h0=(u32)h, h1=(u32)(h>>32), n0=(u32)nq, n1=(u32)(nq>>32);
  u64 E = (u64)x0*w0 + (u64)h0*n0;
  u64 M = (u64)x0*w1 + (u64)x1*w0 + (u64)h0*n1 + (u64)h1*n0;
  asm("" : "+v"(M));
  u32 hi = (u32)(E>>32) + (u32)M;
  return ((u64)hi<<32) | (u32)E;
}
// variant G ("split Shoup"): x*w = x0*W0 + x1*W1 with W1 = w*2^32 mod q and 32-bit quotients Q0, Q1:
// r = x0*W0 + x1*W1 - (mulhi(x0,Q0) + mulhi(x1,Q1)) * q in [0, 4q); two mad chains, no carries, no pair building.
struct MulOp2 { u64 w0, w1; u32 q0, q1; };
__device__ __forceinline__ u64 mulG(u64 x, u64 W0, u64 W1, u32 Q0, u32 Q1, u64 nq) {
  u32 x0=(u32)x, x1=(u32)(x>>32), n0=(u32)nq, n1=(u32)(nq>>32);
  u32 h0 = __umulhi(x0, Q0), h1 = __umulhi(x1, Q1);
  u64 E = (u64)x0*(u32)W0 + (u64)x1*(u32)W1 + (u64)h0*n0 + (u64)h1*n0;
  u64 M = (u64)x0*(u32)(W0>>32) + (u64)x1*(u32)(W1>>32) + (u64)h0*n1 + (u64)h1*n1;
  asm("" : "+v"(M));
  u32 hi = (u32)(E>>32) + (u32)M;
  return ((u64)hi<<32) | (u32)E;
}
// variant H: variant G written as one asm block (two interleaved mad chains)
__device__ __forceinline__ u64 mulH(u64 x, u64 W0, u64 W1, u32 Q0, u32 Q1, u64 nq) {
  u64 E, M; u32 h0, h1;
  asm("v_mul_hi_u32 %2, %4, %10\n\t"
      "v_mul_hi_u32 %3, %5, %11\n\t"
      "v_mad_u64_u32 %0, vcc, %4, %6, 0\n\t"
      "v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %8, %0\n\t"
      "v_mad_u64_u32 %1, vcc, %5, %9, %1\n\t"
      "v_mad_u64_u32 %0, vcc, %2, %12, %0\n\t"
      "v_mad_u64_u32 %1, vcc, %2, %13, %1\n\t"
      "v_mad_u64_u32 %0, vcc, %3, %12, %0\n\t"
      "v_mad_u64_u32 %1, vcc, %3, %13, %1"
      : "=&v"(E), "=&v"(M), "=&v"(h0), "=&v"(h1)
      : "v"((u32)x), "v"((u32)(x>>32)), "v"((u32)W0), "v"((u32)(W0>>32)), "v"((u32)W1), "v"((u32)(W1>>32)),
        "v"(Q0), "v"(Q1), "v"((u32)nq), "v"((u32)(nq>>32))
      : "vcc");
  u32 hi = (u32)(E>>32) + (u32)M;
  return ((u64)hi<<32) | (u32)E;
}

template<int V> __global__ __launch_bounds__(256) void bfly(u64* out, const MulOp* tw, u64 q, int iters, long long* stamps){
  long long c0 = clock64(), r0 = wall_clock64();
  u64 x[16]; const u64 q2=q<<1, nq=0-q, nq2=0-q2;
  for(int i=0;i<16;++i) x[i]=out[(size_t)(blockIdx.x*256+threadIdx.x)*16+i];
  for(int it=0;it<iters;++it){
    MulOp w = tw[(it&15)*256+threadIdx.x];
#pragma unroll
    for(int j=0;j<8;++j){
      u64 u, v;
      if(V==0){ u = x[j]>=q2? x[j]-q2: x[j]; v = mulA(x[j+8], w, q); x[j]=u+v; x[j+8]=u+q2-v; }
      else if(V==1){ u64 t = add64(x[j], nq2); u = (int)(t>>32) < 0 ? x[j] : t; v = mulD(x[j+8], w, nq); x[j]=add64(u,v); x[j+8]=add64(u,q2)-v; }
      else if(V==6){ const u64 q4=q<<2, nq4=0-q4; u64 t = x[j]+nq4; u = x[j]>=q4 ? t : x[j];
                     v = mulG(x[j+8], w.op, w.quo, (u32)(w.quo>>7), (u32)(w.op>>9), nq); x[j]=u+v; x[j+8]=u+q4-v; } // timing only: fake split constants
      else if(V==7){ const u64 q4=q<<2, nq4=0-q4; u64 t = x[j]+nq4; u = x[j]>=q4 ? t : x[j];
                     v = mulH(x[j+8], w.op, w.quo, (u32)(w.quo>>7), (u32)(w.op>>9), nq); x[j]=u+v; x[j+8]=u+q4-v; }
      else if(V==4){ u = x[j]; v = mulA(x[j+8], w, q); x[j]=u+v; x[j+8]=u+q2-v; } // sensitivity: no guard (not exact)
      else if(V==5){ v = mulA(x[j+8], w, q); x[j]=v; x[j+8]=v+q2; }                 // sensitivity: multiply only
      else if(V==3){ u64 t = x[j]+nq2; u = (int)(t>>32) < 0 ? x[j] : t; v = mulF(x[j+8], w, nq); x[j]=u+v; x[j+8]=u+q2-v; }
      else { u64 t = x[j]+nq2; u = (int)(t>>32) < 0 ? x[j] : t; v = mulE(x[j+8], w, nq); x[j]=u+v; x[j+8]=u+q2-v; }
    }
    // rotate roles so that every value passes through both legs
    u64 t0=x[0];
#pragma unroll
    for(int i=0;i<15;++i) x[i]=x[i+1];
    x[15]=t0;
  }
  for(int i=0;i<16;++i) out[(size_t)(blockIdx.x*256+threadIdx.x)*16+i]=x[i];
  if(threadIdx.x==0 && blockIdx.x==7){ stamps[0]=clock64()-c0; stamps[1]=wall_clock64()-r0; }
}
int main(){
  const u64 q = 1152921504606584833ULL;
  const int blocks = 256*8, iters = 4000;
  u64* buf; MulOp* tw; long long* stamps; CK(hipMalloc(&stamps, 16));
  CK(hipMalloc(&buf, (size_t)blocks*256*16*8)); CK(hipMalloc(&tw, 16*256*sizeof(MulOp)));
  std::vector<u64> h((size_t)blocks*256*16); u64 s=88172645463325252ULL;
  for(auto& v: h){ s^=s<<13; s^=s>>7; s^=s<<17; v = s % q; }
  std::vector<MulOp> ht(16*256);
  for(auto& w: ht){ s^=s<<13; s^=s>>7; s^=s<<17; w.op = s % q; w.quo = (u64)((((unsigned __int128)w.op)<<64)/q); }
  CK(hipMemcpy(tw, ht.data(), ht.size()*sizeof(MulOp), hipMemcpyHostToDevice));
  std::vector<u64> res[8];
  for(int v=0; v<8; ++v){
    CK(hipMemcpy(buf, h.data(), h.size()*8, hipMemcpyHostToDevice));
    hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto launch=[&](int it){ if(v==0) hipLaunchKernelGGL(bfly<0>, dim3(blocks), dim3(256), 0, 0, buf, tw, q, it, stamps);
                             else if(v==1) hipLaunchKernelGGL(bfly<1>, dim3(blocks), dim3(256), 0, 0, buf, tw, q, it, stamps);
                             else if(v==6) hipLaunchKernelGGL(bfly<6>, dim3(blocks), dim3(256), 0, 0, buf, tw, q, it, stamps);
                             else if(v==7) hipLaunchKernelGGL(bfly<7>, dim3(blocks), dim3(256), 0, 0, buf, tw, q, it, stamps);
                             else if(v==4) hipLaunchKernelGGL(bfly<4>, dim3(blocks), dim3(256), 0, 0, buf, tw, q, it, stamps);
                             else if(v==5) hipLaunchKernelGGL(bfly<5>, dim3(blocks), dim3(256), 0, 0, buf, tw, q, it, stamps);
                             else if(v==3) hipLaunchKernelGGL(bfly<3>, dim3(blocks), dim3(256), 0, 0, buf, tw, q, it, stamps);
                             else hipLaunchKernelGGL(bfly<2>, dim3(blocks), dim3(256), 0, 0, buf, tw, q, it, stamps); };
    launch(iters); CK(hipDeviceSynchronize());
    CK(hipMemcpy(buf, h.data(), h.size()*8, hipMemcpyHostToDevice));
    CK(hipEventRecord(a)); launch(iters); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms,a,b));
    res[v].resize(h.size()); CK(hipMemcpy(res[v].data(), buf, h.size()*8, hipMemcpyDeviceToHost));
    long long hs[2]; CK(hipMemcpy(hs, stamps, 16, hipMemcpyDeviceToHost));
    double nb=(double)blocks*256*8*iters;
    printf("  in-kernel clock %.0f MHz  ", (double)hs[0]/(double)hs[1]*100.0);
    printf("variant %d: %.3f ms  %.1f Gbfly/s  -> cfg2 forward NTT VALU floor %.2f ms  same_as_v0=%d same_as_v6=%d\n", v, ms, nb/ms/1e6, 17.18e9/(nb/ms*1e3)*1e3, (int)(res[v]==res[0]), (int)(v>=6 && res[v]==res[6]));
  }
  return 0;
}
