// oracle/ref_graph.cc -- TEST INFRASTRUCTURE, not product code.  Second translation unit of oracle/_ref/refdrv.
//
// The graph half of buildReadQGraph48 (paths/long/BuildReadQGraph48.cc:1636,1664) sits in translation units that do
// not compile here (paths/HyperBasevector.h pulls paths/KmerPathInterval.h, which neither compiler in the image
// accepts: DESIGN.md section 2).  What DOES compile in place -- with clang's -fdelayed-template-parsing, the flag
// oracle/build_ref.sh already uses for Vec.cc -- is what HyperBasevector is made of: digraphE<basevector> and
// digraphEX<basevector> (graph/Digraph.h, graph/DigraphTemplate.h: AddEdge's sorted adjacency inserts, ToLeft/ToRight,
// writeBinary), BaseVec (canonical form, reverse complement, operator<, the SwitchHitter iterator), FNV1a,
// vecbvec::WriteAll and BinaryWriter.  This file calls those for every byte that reaches a file and restates only
// the glue, in its own words:
//   paths/long/HBVFromEdges.cc:106-111,132-149,170-238,244-296   vertex discovery, canonical edge order, the
//                                                                 queue-ordered numbering of vertices and edges
//   paths/HyperBasevector.cc:121-125,133-137,668-680              K | digraphE ; K | digraphEX ; Involution
//   10X/WriteFiles.cc:69-101                                      which files a.<K>/ holds
// Row f-2 (read pathing) ends here too: ref_driver.cc's Pather glue leaves every read's path parts in parts.bin (made with
// the real KmerDict / KMer / CF<K>::isRC / bvec iterators); this file edits them into ReadPaths on the REAL digraphE
// (ToLeft/ToRight, To/From, ToEdgeObj/FromEdgeObj, ToSize/FromSize, EdgeObject) and writes a.paths with the real
// ReadPath / ReadPathVec feudal writer (paths/long/ReadPath.h, compiled in place).  Restated glue, in its own words:
//   paths/long/BuildReadQGraph48.cc:1212-1317   HBVPather::algorithmTwo (what StageBuildGraph selects: useNewAligner = True,
//                                               10X/runstages/RunStages.cc:389-390)
//   paths/long/BuildReadQGraph48.cc:657-666,796-803,1365-1402   isConformingCapturedGap, isJoinable, pathPartsToReadPath
//   paths/long/ExtendReadPath.cc:15-333         the two overlap scorers and the left / right extension
// Row f-4 follows on the same ReadPathVec: the inverted paths index and the duplicate marks, written by the real
// IncrementalWriter<ULongVec> (feudal/IncrementalWriter.h) and BinaryWriter; restated glue:
//   10X/PathsIndex.cc:23-146      writePathsIndex: (edge, read) pairs sorted, one list of read ids per edge in edge order
//                                 (a.paths.inv); reads per edge, summed with the involuted edge's (a.countsb).  The
//                                 reference walks the edges in PI_CHUNKS = 30 chunks and overruns when the graph has fewer
//                                 than ~870 edges (`e != emax` with i * chunk_size > num_edges, SURVEY 8c caveat 1); what it
//                                 writes when it does not crash is what is written here, for any number of edges.
//   10X/SecretOps.cc:410-566      MarkDups: reads with the same first edge, offset and first five bases of the MATE are
//                                 duplicates of each other; the one whose pair has the highest quality sum (the lowest
//                                 read id among equals) stays, the others' pairs are marked (a.dup)
// Input: the canonical unipath edges (edges.fastb, written by ref_driver.cc's buildEdges glue through the real Dict).
#include "Basevector.h"
#include "feudal/BinaryStream.h"
#include "graph/Digraph.h"
#include "graph/DigraphTemplate.h"
#include "paths/long/ReadPath.h"
#include "feudal/IncrementalWriter.h"
#include "Intvector.h"
#include "feudal/PQVec.h"
#include "Qualvector.h"
#include "math/Hash.h"
#include "system/System.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <deque>
#include <map>
#include <string>
#include <vector>

namespace {

struct EndKey {                                  // one (K-1)-mer at an end of an edge, read in one orientation
    bvec const* bv; bool rc; unsigned pos, klo;
    bvec::SwitchHitterIter begin() const { return bvec::SwitchHitterIter(bv, pos, rc); }
    bvec::SwitchHitterIter end() const { return begin() + klo; }
};

struct IO { int edge; bool rc; };
struct Vtx { int id = -1; std::vector<IO> inc; };

// length descending, then lexical (HBVFromEdges.cc:106-111)
bool edgeLess( bvec const& a, bvec const& b ) { return a.size() != b.size() ? a.size() > b.size() : a < b; }


// ------------------------------------------------------------------------------------------------ row f-2 glue
struct Part                                       // as ref_driver.cc wrote it: elen == 0 is a gap
{   uint32_t edge; int32_t off; uint32_t len, elen;
    bool gap() const { return elen == 0; }
    bool rc() const { return off < 0; }
    unsigned offset() const { return off < 0 ? ~off : off; }
    unsigned endOffset() const { return offset()+len; }
    bool sameEdge( Part const& o ) const { return edge == o.edge && rc() == o.rc(); }
    static Part gapOf( unsigned n ) { return Part{~0u,0,n,0u}; } };

struct PathGlue
{
    digraphE<basevector> const& g; vecbvec const& canon; vec<int> const& fwd; vec<int> const& rev; unsigned K;
    vec<int> toLeft, toRight;
    static unsigned const JITTER = 3;             // HBVPather::MAX_JITTER (:1404)
    mutable size_t hit[16] = {};                  // how often each rule fired (printed: the fixtures must reach every one)

    PathGlue( digraphE<basevector> const& g_, vecbvec const& c, vec<int> const& f, vec<int> const& r, unsigned k )
    : g(g_), canon(c), fwd(f), rev(r), K(k) { g.ToLeft(toLeft); g.ToRight(toRight); }

    int hbvEdge( Part const& p ) const { return p.rc() ? rev[p.edge] : fwd[p.edge]; }
    int kmersOf( int e ) const { return int(g.EdgeObject(e).size())-int(K)+1; }

    // PathPart::isConformingCapturedGap (:657-666): all in unsigned, then through int
    bool conforming( Part const& before, Part const& gap, Part const& after ) const
    { unsigned dist = after.offset()-before.endOffset();
      if ( !before.sameEdge(after) ) dist += before.elen;
      return unsigned(std::abs(int(gap.len-dist))) <= JITTER; }

    // Pather::isJoinable (:796-803): the last K-1 bases of the first part's edge, as the read runs along it, against the
    // first K-1 of the second's
    bool joinable( Part const& a, Part const& b ) const
    { if ( a.edge == b.edge ) return true;
      bvec const& e1 = canon[a.edge]; bvec const& e2 = canon[b.edge];
      unsigned const klo = K-1;
      for ( unsigned i = 0; i != klo; ++i )
      { unsigned char x = a.rc() ? *(e1.rcbegin(e1.size()-klo+i)) : e1[e1.size()-klo+i];
        unsigned char y = b.rc() ? *(e2.rcbegin(i)) : e2[i];
        if ( x != y ) return false; }
      return true; }

    // pathPartsToReadPath (:1365-1402)
    void toReadPath( std::vector<Part> const& parts, ReadPath& path ) const
    { path.clear();
      Part const* last = nullptr;
      for ( Part const& p : parts )
      { if ( p.gap() ) continue;
        if ( last && last->sameEdge(p) ) continue;
        path.push_back(hbvEdge(p)); last = &p; }
      if ( path.empty() ) path.setOffset(0);
      else if ( !parts.front().gap() ) path.setOffset(parts.front().offset());
      else path.setOffset(int(parts[1].offset())-int(parts.front().len)); }

    // scoreLeftOverlap / scoreRightOverlap (ExtendReadPath.cc:15-113): mismatches cost the base's quality (Q2 counts as
    // 20) plus a running penalty that a matching base shrinks by a fifth -- `unsigned -= double`, i.e. truncation of
    // penalty - 0.2*penalty in double; read bases left over beyond the edge cost 10 each
    static unsigned score( bvec const& read, qvec const& q, size_t start, bvec const& edge, unsigned K, bool left )
    { double const decay = 0.2;
      unsigned sum = 0, penalty = 0;
      long r = left ? long(start)-1 : long(read.size())-long(start);      // read index, moving outwards
      long e = left ? long(edge.size())-long(K) : long(K)-1;              // edge index beside the shared K-1 bases
      long const step = left ? -1 : 1;
      auto readIn = [&]() { return r >= 0 && r < long(read.size()); };
      auto edgeIn = [&]() { return e >= 0 && e < long(edge.size()); };
      while ( readIn() && edgeIn() )
      { if ( read[r] != edge[e] ) { int cost = q[r] == 2 ? 20 : int(q[r]); penalty += cost; sum += penalty; }
        else if ( penalty > 0 ) penalty -= (decay*penalty);
        r += step; e += step; }
      while ( readIn() ) { sum += 10; r += step; }
      return sum; }

    // attemptLeftwardExtension / attemptRightwardExtension (ExtendReadPath.cc:130-244, 247-378)
    bool extend( ReadPath& path, bvec const& read, qvec const& q, bool left ) const
    { if ( !path.size() ) return false;
      size_t hang;
      if ( left )
      { if ( path.getOffset() >= 0 ) return false;
        hang = -path.getOffset(); }
      else
      { int h = read.size(); h += path.getOffset();
        for ( int e : path ) h -= kmersOf(e);
        h -= int(K)-1;
        if ( h < 10 ) return false;
        hang = h; }
      if ( hang < 10 ) return false;
      int const v = left ? toLeft[path.front()] : toRight[path.back()];
      vec<int> const& cand = left ? g.ToEdgeObj(v) : g.FromEdgeObj(v);
      vec<int> const& far = left ? g.To(v) : g.From(v);
      auto deadEnd = [&]( int w ) { return left ? ( g.ToSize(w) == 0 && g.FromSize(w) == 1 ) : ( g.FromSize(w) == 0 && g.ToSize(w) == 1 ); };
      std::vector<bool> hanging(cand.size(),false), reaches(cand.size(),false);
      std::vector<int> shortTo;
      size_t nReach = 0;
      for ( size_t i = 0; i != cand.size(); ++i )
      { hanging[i] = deadEnd(far[i]);
        reaches[i] = g.EdgeObject(cand[i]).size()-(K-1) >= hang;
        nReach += reaches[i];
        if ( !reaches[i] && !hanging[i] ) shortTo.push_back(far[i]); }
      if ( cand.size() != 1 && !shortTo.empty() )
      { ++hit[12]; if ( nReach ) return false;
        std::sort(shortTo.begin(),shortTo.end()); shortTo.erase(std::unique(shortTo.begin(),shortTo.end()),shortTo.end());
        if ( shortTo.size() != 1 ) return false;
        if ( ( left ? g.ToSize(shortTo.back()) : g.FromSize(shortTo.back()) ) != 1 ) return false; ++hit[13]; }
      int best = -1; unsigned least = std::numeric_limits<unsigned>::max();
      for ( size_t i = 0; i != cand.size(); ++i )
        if ( !hanging[i] || cand.size() == 1 )
        { unsigned s = score(read,q,hang,g.EdgeObject(cand[i]),K,left);
          if ( s < least ) { least = s; best = cand[i]; } }
      if ( best == -1 || least > hang*10 ) { ++hit[ best == -1 ? 14 : 15 ]; return false; }
      if ( left )
      { ReadPath longer; longer.setOffset(path.getOffset()+kmersOf(best));
        longer.push_back(best);
        for ( int e : path ) longer.push_back(e);
        path = longer; }
      else path.push_back(best);
      return true; }

    // HBVPather::algorithmTwo (:1212-1317) on the parts Pather::path produced
    void edit( std::vector<Part> parts, bvec const& read, qvec const& q, ReadPath& path ) const
    { // seeds on short hanging edges become gaps; neighbouring gaps merge
      std::vector<Part> kept;
      for ( Part p : parts )
      { if ( !p.gap() )
        { int e = hbvEdge(p); int vl = toLeft[e], vr = toRight[e];
          if ( g.ToSize(vl) == 0 && g.ToSize(vr) > 1 && g.FromSize(vr) > 0 && p.elen <= 100 ) { p = Part::gapOf(p.len); ++hit[0]; } }
        if ( p.gap() && !kept.empty() && kept.back().gap() ) { kept.back().len += p.len; ++hit[1]; }
        else kept.push_back(p); }
      parts.swap(kept);
      // the first captured gap that the graph does not explain ends the path: with more than one seed before it the seed
      // in front of it goes too
      if ( parts.size() >= 3 )
      { size_t seeds = parts.front().gap() ? 0 : 1;
        for ( size_t i = 1; i+1 < parts.size(); ++i )
        { if ( !parts[i].gap() ) { ++seeds; continue; }
          if ( conforming(parts[i-1],parts[i],parts[i+1]) && joinable(parts[i-1],parts[i+1]) ) { ++hit[2]; continue; }
          ++hit[ conforming(parts[i-1],parts[i],parts[i+1]) ? 3 : 4 ];
          if ( seeds > 1 )
          { ++hit[5]; Part tail = Part::gapOf(parts[i-1].len);
            for ( size_t j = i; j != parts.size(); ++j ) tail.len += parts[j].len;
            parts.resize(i-1); parts.push_back(tail); }
          else
          { ++hit[6]; for ( size_t j = i+1; j != parts.size(); ++j ) parts[i].len += parts[j].len;
            parts.resize(i+1); }
          break; } }
      // a last seed of at most five k-mers at the very start of its edge is not trusted
      if ( parts.back().gap() && parts.size() > 1 )
      { Part const& seed = parts[parts.size()-2];
        if ( seed.offset() == 0 && seed.len <= 5 )
        { ++hit[7]; Part merged = parts.back(); merged.len += seed.len;
          parts.pop_back(); parts.pop_back(); parts.push_back(merged); } }
      else if ( !parts.back().gap() )
      { Part& seed = parts.back();
        if ( seed.offset() == 0 && seed.len <= 5 ) { seed = Part::gapOf(seed.len); ++hit[8]; } }
      toReadPath(parts,path);
      // consecutive edges must meet at a vertex
      for ( size_t i = 0; i+1 < path.size(); ++i )
        if ( toRight[path[i]] != toLeft[path[i+1]] ) { path.resize(i+1); ++hit[9]; break; }
      while ( extend(path,read,q,true) ) ++hit[10];
      while ( extend(path,read,q,false) ) ++hit[11]; }
};

int paths_main( digraphE<basevector> const& g, vecbvec const& canon, vec<int> const& fwd, vec<int> const& rev, unsigned K,
                std::string const& readsHead, std::string const& partsFile, std::string const& dir, vec<int> const* pInv )
{
    vecbvec reads; reads.ReadAll((readsHead+".fastb").c_str());
    VecPQVec quals; quals.ReadAll((readsHead+".qualp").c_str());
    std::ifstream in(partsFile.c_str(),std::ios::binary);
    uint64_t n = 0; in.read((char*)&n,8);
    if ( n != reads.size() ) { fprintf(stderr,"parts.bin holds %lu reads, the input %lu\n",(unsigned long)n,(unsigned long)reads.size()); return 2; }
    PathGlue pg(g,canon,fwd,rev,K);
    ReadPathVec paths; paths.reserve(n);
    std::vector<Part> parts; qvec q; ReadPath path;
    size_t placed = 0, nEdges = 0;
    for ( uint64_t r = 0; r != n; ++r )
    { uint32_t m = 0; in.read((char*)&m,4); parts.resize(m); in.read((char*)parts.data(),16*size_t(m));
      quals[r].unpack(&q);
      pg.edit(parts,reads[r],q,path);
      placed += path.size() != 0; nEdges += path.size();
      paths.push_back(path); }
    paths.WriteAll((dir+"/a.paths").c_str());                               // WriteFiles.cc:78-82
    {   // ---- writePathsIndex (10X/PathsIndex.cc:23-146), as DF.cc:550 calls it on the paths StageBuildGraph returned
        int const nEdges = g.EdgeObjectCount();
        std::vector<std::pair<int,unsigned long>> where;
        for ( uint64_t id = 0; id != n; ++id ) for ( int e : paths[id] ) where.push_back(std::make_pair(e,(unsigned long)id));
        std::sort(where.begin(),where.end());
        vec<vec<int>> counts(1,vec<int>(nEdges,0));
        {   IncrementalWriter<ULongVec> w((dir+"/a.paths.inv").c_str());
            size_t j = 0;
            for ( int e = 0; e != nEdges; ++e )
            { ULongVec ids;
              while ( j != where.size() && where[j].first == e ) ids.push_back(where[j++].second);
              w.add(ids); counts[0][e] = ids.size(); }
            w.close(); }
        vec<int> const& inv = *pInv;
        for ( int e = 0; e != nEdges; ++e )
          if ( e < inv[e] ) { int both = counts[0][e]+counts[0][inv[e]]; counts[0][e] = both; counts[0][inv[e]] = both; }
        BinaryWriter::writeFile((dir+"/a.countsb").c_str(),counts); }
    {   // ---- MarkDups (10X/SecretOps.cc:410-566); `art`, the share of artifactual duplicates, is a statistic and not a file
        struct X { int e, off, head; int64_t id; };
        std::vector<X> xs(n);
        for ( uint64_t id = 0; id != n; ++id )
        { uint64_t mate = id^1ul;
          if ( paths[id].size() == 0 ) { xs[id] = X{-1,-1,-1,-1}; continue; }
          int head = 0; for ( int j = 0; j != 5; ++j ) head = 4*head+reads[mate][j];
          xs[id] = X{paths[id][0],paths[id].getOffset(),head,int64_t(id)}; }
        std::sort(xs.begin(),xs.end(),[]( X const& a, X const& b )
          { if ( a.e != b.e ) return a.e < b.e; if ( a.off != b.off ) return a.off < b.off; if ( a.head != b.head ) return a.head < b.head; return a.id < b.id; });
        auto qsumOf = [&]( int64_t id ) { int s = 0; quals[id].unpack(&q); for ( unsigned char v : q ) s += v; quals[id^1].unpack(&q); for ( unsigned char v : q ) s += v; return s; };
        vec<Bool> dup(n/2,False);
        size_t groups = 0;
        for ( size_t j = 0; j < xs.size(); )
        { size_t k = j+1;
          while ( k < xs.size() && xs[k].e == xs[j].e && xs[k].off == xs[j].off && xs[k].head == xs[j].head ) ++k;
          if ( xs[j].e >= 0 && k-j > 1 )
          { ++groups;
            size_t best = j; int top = qsumOf(xs[j].id);
            for ( size_t l = j+1; l != k; ++l ) { int s = qsumOf(xs[l].id); if ( s > top ) { top = s; best = l; } }     // ties: the earlier (lower id) stays
            for ( size_t l = j; l != k; ++l ) if ( l != best ) dup[xs[l].id/2] = True; }
          j = k; }
        BinaryWriter::writeFile((dir+"/a.dup").c_str(),dup);
        size_t nd = 0; for ( Bool b : dup ) nd += b;
        printf("dups: %zu groups, %zu of %lu pairs marked\n",groups,nd,(unsigned long)(n/2)); }
    printf("paths: %lu reads, %zu placed, %zu path edges\n",(unsigned long)n,placed,nEdges);
    static char const* what[16] = { "seed on a hanging edge dropped", "gaps merged", "captured gap accepted", "captured gap not joinable", "captured gap not conforming",
        "... cut with the seed before it", "... cut after it", "short last seed before a gap dropped", "short last seed dropped", "path cut at a non-adjacent edge",
        "left extensions", "right extensions", "short-edge rule consulted", "short-edge rule passed", "no candidate edge", "best score too high" };
    for ( int i = 0; i != 16; ++i ) printf("paths:   %-40s %zu\n",what[i],pg.hit[i]);
    return 0;
}

} // namespace

int graph_main( unsigned K, std::string const& edgesFile, std::string const& dir, std::string const& readsHead, std::string const& partsFile )
{
    vecbvec edges; edges.ReadAll(edgesFile.c_str());
    size_t const nE = edges.size();
    unsigned const klo = K - 1;
    Mkdir777(dir.c_str());
    digraphE<basevector> g;
    vec<int> fwd(nE, -1), rev(nE, -1);
    if ( nE )
    {
        // vertices = distinct (K-1)-mers at the edge ends, in both orientations (a palindromic edge has one)
        std::map<std::vector<unsigned char>, Vtx> verts;
        auto keyOf = [&]( EndKey const& e ) { std::vector<unsigned char> k; for ( auto i = e.begin(), z = e.end(); i != z; ++i ) k.push_back(*i); return k; };
        auto endOf = [&]( size_t id, bool rc, bool distal ) { bvec const& b = edges[id]; return EndKey{&b, rc, distal ? unsigned(b.size()) - klo : 0u, klo}; };
        // canonical edge order first: a vertex lists its incident (edge, orientation, end) by that order, then
        // forward before reverse, then near end before far end (EEComp, :113-121)
        std::vector<size_t> order(nE);
        for ( size_t i = 0; i != nE; ++i ) order[i] = i;
        std::sort(order.begin(), order.end(), [&]( size_t a, size_t b ) { return edgeLess(edges[a], edges[b]); });
        for ( size_t id : order )
        {
            bool pal = edges[id].getCanonicalForm() == CanonicalForm::PALINDROME;
            for ( int rc = 0; rc < (pal ? 1 : 2); ++rc )
                for ( int distal = 0; distal < 2; ++distal )
                    verts[keyOf(endOf(id, rc, distal))].inc.push_back(IO{int(id), bool(rc)});
        }
        g.AddVertices(int(verts.size()));
        int nextV = 0;
        std::deque<IO> q;
        auto done = [&]( IO x ) { return (x.rc ? rev : fwd)[x.edge] != -1; };
        auto add = [&]( IO first )
        {
            if ( done(first) ) return;
            q.push_back(first);
            while ( !q.empty() )
            {
                IO x = q.front(); q.pop_front();
                if ( done(x) ) continue;
                bvec const& e = edges[x.edge];
                Vtx& a = verts[keyOf(endOf(x.edge, x.rc, false))];
                if ( a.id < 0 ) a.id = nextV++;
                Vtx& b = verts[keyOf(endOf(x.edge, x.rc, true))];
                if ( b.id < 0 ) b.id = nextV++;
                int newId = g.EdgeObjectCount();
                g.AddEdge(a.id, b.id, e);                            // the reference's sorted adjacency insert
                if ( x.rc ) g.EdgeObjectMutable(newId).ReverseComplement();
                bool pal = e.getCanonicalForm() == CanonicalForm::PALINDROME;
                if ( !x.rc || pal ) fwd[x.edge] = newId;
                if ( x.rc || pal ) rev[x.edge] = newId;
                for ( IO y : a.inc ) if ( !done(y) ) q.push_back(y);
                for ( IO y : b.inc ) if ( !done(y) ) q.push_back(y);
            }
        };
        for ( size_t id : order ) add(IO{int(id), false});
        for ( size_t id : order ) add(IO{int(id), true});
    }
    // a.<K>/ (WriteFiles.cc:69-101).  a.hbv = K | digraphE<basevector> (HyperBasevector.cc:121-125)
    { FILE* f = fopen((dir + "/a.k").c_str(), "w"); fprintf(f, "%u\n", K); fclose(f); }
    { BinaryWriter w((dir + "/a.hbv").c_str()); int k = int(K); w.write(k); w.write(g); }
    vec<int> toLeft, toRight; g.ToLeft(toLeft); g.ToRight(toRight);
    BinaryWriter::writeFile((dir + "/a.to_left").c_str(), toLeft);
    BinaryWriter::writeFile((dir + "/a.to_right").c_str(), toRight);
    // Involution (HyperBasevector.cc:668-680): rank the edges, rank their reverse complements, match rank to rank
    int const E = g.EdgeObjectCount();
    vec<int> inv(E);
    {
        vecbvec es(E);
        for ( int e = 0; e < E; ++e ) es[e] = g.EdgeObject(e);
        std::vector<int> x1(E), x2(E);
        for ( int e = 0; e < E; ++e ) x1[e] = x2[e] = e;
        std::sort(x1.begin(), x1.end(), [&]( int a, int b ) { return es[a] < es[b]; });
        for ( int e = 0; e < E; ++e ) es[e].ReverseComplement();
        std::sort(x2.begin(), x2.end(), [&]( int a, int b ) { return es[a] < es[b]; });
        for ( int i = 0; i < E; ++i ) inv[x1[i]] = x2[i];
    }
    BinaryWriter::writeFile((dir + "/a.inv").c_str(), inv);
    { BinaryWriter w((dir + "/a.hbx").c_str()); int k = int(K); w.write(k); digraphEX<basevector> gx(g); w.write(gx); }   // HyperBasevectorX (HyperBasevector.cc:133-137)
    { vecbvec out(g.Edges().begin(), g.Edges().end()); out.WriteAll((dir + "/a.fastb").c_str()); }
    { vec<int> kmers(E); for ( int e = 0; e < E; ++e ) kmers[e] = int(g.EdgeObject(e).size()) - int(K) + 1; BinaryWriter::writeFile((dir + "/a.kmers").c_str(), kmers); }
    // the translation tables pathReads uses (canonical edge -> HBV edge, both orientations)
    BinaryWriter::writeFile((dir + "/fwd_xlat").c_str(), fwd);
    BinaryWriter::writeFile((dir + "/rev_xlat").c_str(), rev);
    printf("graph: %zu canonical edges -> %d vertices, %d edges\n", nE, g.N(), E);
    if ( !readsHead.empty() ) return paths_main(g, edges, fwd, rev, K, readsHead, partsFile, dir, &inv);
    return 0;
}
