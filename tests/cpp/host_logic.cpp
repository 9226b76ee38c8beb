// Host-side logic of the drop-in headers that needs no GPU (compiled and run by tests/test_host_cpp.py):
// partial PROSAC order == prefix of the full order (ties included), lazy extension, cvtInlier snapshot semantics,
// sparse RandomElements == the reference's re-initialised Fisher-Yates table.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "AOOnlyPoseAdapter.hpp"
#include "NormalAOPoseAdapter.hpp"
#include "Utility.hpp"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main() {
  {  // ---- content fingerprint of a host array (rpe::DeviceSet keys its resident copies by address AND content)
    std::vector<float> a(3 * 307200);
    for (size_t i = 0; i < a.size(); i++) a[i] = (float)(i % 1013) * 0.25f;
    const size_t bytes = a.size() * sizeof(float);
    const unsigned long long f0 = rpe::host_fingerprint(a.data(), bytes);
    CHECK(f0 == rpe::host_fingerprint(a.data(), bytes));                    // a pure function of the content
    std::vector<float> b(a);
    CHECK(f0 == rpe::host_fingerprint(b.data(), bytes));                    // ... not of the address
    b[0] += 1.0f;
    CHECK(f0 != rpe::host_fingerprint(b.data(), bytes));                    // the first line is sampled
    b = a; b.back() += 1.0f;
    CHECK(f0 != rpe::host_fingerprint(b.data(), bytes));                    // and the last
    CHECK(f0 != rpe::host_fingerprint(a.data(), bytes - sizeof(float)));    // the length enters
    for (float& x : b) x += 0.5f;                                           // a refilled buffer (every line changes) is always caught
    CHECK(f0 != rpe::host_fingerprint(b.data(), bytes));
    CHECK(rpe::host_fingerprint(nullptr, 0) == rpe::host_fingerprint(a.data(), 0));
    {  // FULL mode: every byte enters -- a sparse in-place edit (three columns NaN-marked, the reference's idiom) changes the value
      std::vector<float> c(a);
      const unsigned long long s0 = rpe::host_fingerprint(c.data(), bytes), g0 = rpe::host_fingerprint(c.data(), bytes, true);
      CHECK(g0 == rpe::host_fingerprint(a.data(), bytes, true) && g0 != s0);
      for (int col : {1234, 100001, 250000}) for (int k = 0; k < 3; k++) c[3 * (size_t)col + k] = NAN;
      CHECK(rpe::host_fingerprint(c.data(), bytes) == s0);                  // the sampled mode does not see it (documented)
      CHECK(rpe::host_fingerprint(c.data(), bytes, true) != g0);            // the full mode does
      for (size_t off : {(size_t)0, (size_t)7, bytes / 2 + 3, bytes - 1}) {  // any single byte, wherever it sits (tails included)
        std::vector<float> d(a);
        reinterpret_cast<unsigned char*>(d.data())[off] ^= 0x40;
        CHECK(rpe::host_fingerprint(d.data(), bytes, true) != g0);
        CHECK(rpe::host_fingerprint(d.data(), bytes - 5, true) != rpe::host_fingerprint(a.data(), bytes - 5, true) || off >= bytes - 5);
      }
    }
    float tiny[3] = {1.f, 2.f, 3.f};                                        // arrays shorter than a cache line
    const unsigned long long ft = rpe::host_fingerprint(tiny, sizeof tiny);
    tiny[2] = 4.f;
    CHECK(ft != rpe::host_fingerprint(tiny, sizeof tiny));
  }
  // ---- sortIndexes: partial prefix, ties to the lower index
  rpe::Rand31 rnd(3);
  std::vector<float> w(5000);
  for (float& x : w) x = (float)(rnd() % 97) / 7.0f;   // many ties
  const std::vector<int> full = sortIndexes<float>(w);
  for (size_t i = 1; i < full.size(); i++) CHECK(w[full[i - 1]] > w[full[i]] || (w[full[i - 1]] == w[full[i]] && full[i - 1] < full[i]));
  for (int k : {0, 1, 7, 300, 4999, 5000, 9000}) {
    const std::vector<int> part = sortIndexes<float>(w, k);
    CHECK(part.size() == (size_t)std::min(k, 5000));
    for (size_t i = 0; i < part.size(); i++) CHECK(part[i] == full[i]);
  }
  {
    std::vector<int> order = sortIndexes<float>(w, 10), sel = {0, 9, 3};
    mapSortedIdx<float>(w, order, sel);
    CHECK(order.size() == 10 && sel[0] == full[0] && sel[1] == full[9] && sel[2] == full[3]);
    sel = {2, 10, 4321};                         // beyond the prefix: extended to the full order, still exact
    mapSortedIdx<float>(w, order, sel);
    CHECK(order.size() == 5000 && sel[0] == full[2] && sel[1] == full[10] && sel[2] == full[4321]);
  }
  {  // long array, short prefix: the subsampled-cut path, with heavy ties and with all-equal weights (cut keeps everything)
    std::vector<float> big(200000);
    for (float& x : big) x = (float)(rnd() % 1000);
    const std::vector<int> fb = sortIndexes<float>(big);
    for (int k : {1, 100, 3053, 12000}) {
      const std::vector<int> pb = sortIndexes<float>(big, k);
      CHECK(pb.size() == (size_t)k);
      for (size_t i = 0; i < pb.size(); i++) CHECK(pb[i] == fb[i]);
    }
    std::vector<double> flat(100000, 1.0);
    const std::vector<int> pf = sortIndexes<double>(flat, 50);
    for (int i = 0; i < 50; i++) CHECK(pf[i] == i);
    flat[77777] = 2.0;
    CHECK(sortIndexes<double>(flat, 3) == (std::vector<int>{77777, 0, 1}));
  }
  // ---- adapters: sortIdx(top_k) + getSortedIdx, cvtInlier / getInlierIdx
  const int N = 1000;
  rpe::MatrixX<float> P(3, N), Q(3, N), Wt(N, 3);
  for (int i = 0; i < N; i++) { Wt(i, 0) = (float)(rnd() % 50); Wt(i, 1) = (float)(rnd() % 1000); Wt(i, 2) = 1.f; }
  AOOnlyPoseAdapter<float> a(P, Q);
  a.setWeights(Wt);
  std::vector<float> w1(N);
  for (int i = 0; i < N; i++) w1[i] = Wt(i, 1);
  const std::vector<int> f1 = sortIndexes<float>(w1);
  a.sortIdx(20);
  std::vector<int> sel = {0, 19, 5};
  a.getSortedIdx(sel);
  CHECK(sel[0] == f1[0] && sel[1] == f1[19] && sel[2] == f1[5]);
  sel = {999, 20};
  a.getSortedIdx(sel);
  CHECK(sel[0] == f1[999] && sel[1] == f1[20]);
  // the order is cached per weight set: a new setWeights invalidates it
  {
    rpe::MatrixX<float> W2(N, 3);
    for (int i = 0; i < N; i++) { W2(i, 0) = 1.f; W2(i, 1) = (float)(N - i); W2(i, 2) = 1.f; }   // descending: order = identity
    a.sortIdx(10);
    a.setWeights(W2);
    a.sortIdx(10);
    std::vector<int> s2 = {0, 9};
    a.getSortedIdx(s2);
    CHECK(s2[0] == 0 && s2[1] == 9);
    a.sortIdx(5);                                   // shorter prefix than cached: reused
    s2 = {4, 700};                                  // 700 is beyond any prefix: extended to the full order
    a.getSortedIdx(s2);
    CHECK(s2[0] == 4 && s2[1] == 700);
    a.setWeights(Wt);                               // back to the first weights for the checks below
  }
  // cvtInlier lists the mask AS IT WAS when it was called, even though the list is built lazily
  rpe::MatrixXs m(N, 2);
  for (int i = 0; i < N; i++) m(i, 1) = (short)(i % 3 == 0);
  a.setInlier(m);
  a.cvtInlier();
  for (int i = 0; i < N; i++) m(i, 1) = (short)(i % 5 == 0);
  a.setInlier(m);                                 // mask changes AFTER cvtInlier and BEFORE the first read
  const std::vector<int>& idx = a.getInlierIdx();
  CHECK(idx.size() == 334);
  for (size_t i = 0; i < idx.size(); i++) CHECK(idx[i] == (int)(3 * i));
  a.cvtInlier();
  CHECK(a.getInlierIdx().size() == 200 && a.getInlierIdx()[1] == 5);
  a.cvtInlier();
  a.inlierMask33()[0] = 0;                        // the backend's write access also snapshots first
  CHECK(a.getInlierIdx().size() == 200 && a.getInlierIdx()[0] == 0);
  // what a solver does: drop the unread request, replace the mask, request again -> the reader sees the new mask's list
  a.cvtInlier();
  a.forgetInlierIdx();
  for (int i = 0; i < N; i++) m(i, 1) = (short)(i % 100 == 0);
  a.setInlier(m);
  a.cvtInlier();
  CHECK(a.getInlierIdx().size() == 10 && a.getInlierIdx()[9] == 900);
  // setInlierFromDevice with no device column = setInlier of an all-zero matrix (what the reference's solvers leave in the columns
  // they do not vote on); cols == 1 leaves the 3-D mask alone, as setInlier does
  a.setInlierFromDevice(1, 0u);
  CHECK(a.isInlier33(0) && a.isInlier33(100));
  a.setInlierFromDevice(2, 0u);
  a.cvtInlier();
  CHECK(!a.isInlier33(0) && !a.isInlier33(100) && a.getInlierIdx().empty());
  // every level of the NormalAO hierarchy keeps its own list (name hiding, reference AbsoluteOrientation.hpp:433-435)
  rpe::MatrixX<float> U(3, N), Nc(3, N), Nw(3, N);
  NormalAOPoseAdapter<float> na(U, P, Nc, Q, Nw);
  rpe::MatrixXs m3(N, 3);
  for (int i = 0; i < N; i++) { m3(i, 0) = (short)(i % 2 == 0); m3(i, 1) = (short)(i % 4 == 0); m3(i, 2) = (short)(i % 10 == 0); }
  na.setInlier(m3);
  PnPPoseAdapter<float>* p23 = &na; AOPoseAdapter<float>* p33 = &na;
  p23->cvtInlier(); p33->cvtInlier(); na.cvtInlier();
  CHECK(p23->getInlierIdx().size() == 500 && p33->getInlierIdx().size() == 250 && na.getInlierIdx().size() == 100);
  CHECK(p23->getInlierIdx()[3] == 6 && p33->getInlierIdx()[3] == 12 && na.getInlierIdx()[3] == 30);
  // ---- RandomElements: same stream as the reference's table that is rebuilt as the identity on every draw (Utility.hpp:124-156)
  {
    const int n = 50;
    rpe::Rand31 r1(11), r2(11);
    RandomElements<int> re(n);
    for (int draw = 0; draw < 200; draw++) {
      const int mm = 1 + draw % 6;
      std::vector<int> got, table(n), want;
      re.run(mm, &got, r1);
      for (int i = 0; i < n; i++) table[i] = i;
      for (int top = n - 1; top > n - mm - 1; top--) { const int pick = r2() % (top + 1); std::swap(table[pick], table[top]); want.push_back(table[top]); }
      CHECK(got == want);
    }
  }
  std::printf(fails ? "host_logic: %d FAILURES\n" : "host_logic: ok\n", fails);
  return fails ? 1 : 0;
}
