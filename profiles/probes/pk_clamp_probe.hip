#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(const v2f *x, const v2f *y, v2f *o){ v2f a = x[threadIdx.x], c = y[threadIdx.x], r;
 asm volatile("v_pk_fma_f32 %0, %1, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(c));
 o[threadIdx.x] = r; }
int main(){ v2f hx[64], hy[64], ho[64]; for(int i=0;i<64;i++){hx[i]=v2f{0.5f*i, -0.25f*i}; hy[i]=v2f{-1.0f, 0.1f};} hx[5]=v2f{NAN,3.f}; hy[6]=v2f{NAN,-100.f};
 v2f *dx,*dy,*d; hipMalloc(&dx,512); hipMalloc(&dy,512); hipMalloc(&d,512); hipMemcpy(dx,hx,512,hipMemcpyHostToDevice); hipMemcpy(dy,hy,512,hipMemcpyHostToDevice);
 hipLaunchKernelGGL(k,dim3(1),dim3(64),0,0,dx,dy,d); hipMemcpy(ho,d,512,hipMemcpyDeviceToHost);
 for(int i=0;i<8;i++) printf("%g %g | ", ho[i].x, ho[i].y); printf("\n"); return 0; }
