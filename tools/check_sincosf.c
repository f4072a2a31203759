#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef struct { double sign[4]; double hpi_inv, hpi, c0,c1,c2,c3,c4, s1,s2,s3; } sincos_t;
static const sincos_t T[2] = {
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static inline uint32_t asuint(float f){uint32_t u; memcpy(&u,&f,4); return u;}
static inline uint32_t abstop12(float x){ return (asuint(x)>>20)&0x7ff; }
static inline double poly(double x, double x2, const sincos_t*p, int n){
  if((n&1)==0){ double x3=x*x2; double s1=p->s2+x2*p->s3; double x7=x3*x2; double s=x+x3*p->s1; return s+x7*s1; }
  else { double x4=x2*x2; double c2=p->c3+x2*p->c4; double c1=p->c0+x2*p->c1; double x6=x4*x2; double c=c1+x4*p->c2; return c+x6*c2; }
}
static inline double reduce_fast(double x,const sincos_t*p,int*np){ double r=x*p->hpi_inv; int n=((int32_t)r+0x800000)>>24; *np=n; return x-n*p->hpi; }
float my_cosf(float y){ double x=y; int n; const sincos_t*p=&T[0];
  if(abstop12(y)<abstop12(0x1.921FB6p-1f)){ double x2=x*x; if(abstop12(y)<abstop12(0x1p-12f)) return 1.0f; return (float)poly(x,x2,p,1);} 
  x=reduce_fast(x,p,&n); double s=p->sign[n&3]; if(n&2)p=&T[1]; return (float)poly(x*s,x*x,p,n^1); }
float my_sinf(float y){ double x=y; int n; const sincos_t*p=&T[0];
  if(abstop12(y)<abstop12(0x1.921FB6p-1f)){ double s=x*x; if(abstop12(y)<abstop12(0x1p-12f)) return y; return (float)poly(x,s,p,0);} 
  x=reduce_fast(x,p,&n); double s=p->sign[n&3]; if(n&2)p=&T[1]; return (float)poly(x*s,x*x,p,n); }
int main(){ long bad_c=0,bad_s=0,tot=0; float lim=7.0f;
  for(uint32_t u=0; ; u++){ float f; memcpy(&f,&u,4); if(f>lim) break; tot++;
    float c=cosf(f), s=sinf(f); if(asuint(c)!=asuint(my_cosf(f))){ if(bad_c<5)printf("cos %a: %a vs %a\n",f,c,my_cosf(f)); bad_c++;}
    if(asuint(s)!=asuint(my_sinf(f))){ if(bad_s<5)printf("sin %a: %a vs %a\n",f,s,my_sinf(f)); bad_s++;} }
  printf("tot %ld bad_c %ld bad_s %ld\n",tot,bad_c,bad_s); return 0; }
