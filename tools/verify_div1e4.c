#include <stdio.h>
#include <math.h>
#include <stdint.h>
int main(){
  const double c1 = 1e-4; // RN(1/1e4)
  uint64_t bad=0, first=0;
  #pragma omp parallel for reduction(+:bad)
  for (int64_t r=0; r< (1LL<<33); r++){
    double x=(double)r;
    double q0 = x*c1;
    double rem = fma(-q0, 1e4, x);
    double q = fma(rem, c1, q0);
    if (q != x/1e4) { bad++; }
  }
  printf("bad=%llu\n",(unsigned long long)bad);
  // half-integers are irrelevant; also test random large doubles
  return 0;
}
