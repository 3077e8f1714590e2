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
// Input: the canonical unipath edges (edges.fastb, written by ref_driver.cc's buildEdges glue through the real Dict).
#include "Basevector.h"
#include "feudal/BinaryStream.h"
#include "graph/Digraph.h"
#include "graph/DigraphTemplate.h"
#include "math/Hash.h"
#include "system/System.h"
#include <algorithm>
#include <cstdio>
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

} // namespace

int graph_main( unsigned K, std::string const& edgesFile, std::string const& dir )
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
    return 0;
}
