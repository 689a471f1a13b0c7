#pragma once
// compile-only stand-in for boost::iterator_facade (absent here): just enough interface for the reference headers
#include <cstddef>
#include <iterator>
namespace boost {
struct random_access_traversal_tag {};
class iterator_core_access {
 public:
    template <class It> static decltype(auto) dereference(const It& it) { return it.dereference(); }
    template <class It> static void increment(It& it) { it.increment(); }
    template <class It> static void decrement(It& it) { it.decrement(); }
    template <class It> static void advance(It& it, std::ptrdiff_t n) { it.advance(n); }
    template <class It> static std::ptrdiff_t distance_to(const It& a, const It& b) { return a.distance_to(b); }
    template <class It> static bool equal(const It& a, const It& b) { return a.equal(b); }
};
template <class Derived, class Value, class Traversal, class Reference = Value&, class Difference = std::ptrdiff_t>
class iterator_facade {
 public:
    using iterator_category = std::random_access_iterator_tag;
    using value_type = std::remove_cv_t<Value>;
    using reference = Reference;
    using pointer = Value*;
    using difference_type = Difference;
    Derived& self() { return static_cast<Derived&>(*this); }
    const Derived& self() const { return static_cast<const Derived&>(*this); }
    reference operator*() const { return iterator_core_access::dereference(self()); }
    Derived& operator++() { iterator_core_access::increment(self()); return self(); }
    Derived operator++(int) { Derived t(self()); ++*this; return t; }
    Derived& operator--() { iterator_core_access::decrement(self()); return self(); }
    Derived& operator+=(difference_type n) { iterator_core_access::advance(self(), n); return self(); }
    Derived& operator-=(difference_type n) { iterator_core_access::advance(self(), -n); return self(); }
    friend Derived operator+(Derived a, difference_type n) { a += n; return a; }
    friend Derived operator-(Derived a, difference_type n) { a -= n; return a; }
    friend difference_type operator-(const Derived& a, const Derived& b) { return iterator_core_access::distance_to(b, a); }
    friend bool operator==(const Derived& a, const Derived& b) { return iterator_core_access::equal(a, b); }
    friend bool operator!=(const Derived& a, const Derived& b) { return !iterator_core_access::equal(a, b); }
    friend bool operator<(const Derived& a, const Derived& b) { return iterator_core_access::distance_to(a, b) > 0; }
    reference operator[](difference_type n) const { Derived t(self()); t += n; return *t; }
};
}  // namespace boost
