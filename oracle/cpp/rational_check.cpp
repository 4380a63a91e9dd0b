// Arithmetic self-check driver for rational.hpp (TEST INFRASTRUCTURE ONLY): reads "op a b" lines (op in + - * / c g),
// prints the exact result; tests/test_oracle_cpp.py compares with Python's Fraction / math.gcd.
#include <iostream>
#include <string>

#include "rational.hpp"

int main() {
    std::string op, a, b;
    while (std::cin >> op >> a >> b) {
        using oracle::Rat;
        if (op == "g") {  // gcd of two naturals given as "n/1"
            oracle::Nat x, y;
            for (char ch : a) x.mul_add_small(10, (uint64_t)(ch - '0'));
            for (char ch : b) y.mul_add_small(10, (uint64_t)(ch - '0'));
            x.trim();
            y.trim();
            std::cout << oracle::Nat::gcd(x, y).to_string() << "\n";
            continue;
        }
        Rat x = Rat::parse(a), y = Rat::parse(b);
        if (op == "+") std::cout << (x + y).to_string() << "\n";
        else if (op == "-") std::cout << (x - y).to_string() << "\n";
        else if (op == "*") std::cout << (x * y).to_string() << "\n";
        else if (op == "/") std::cout << (x / y).to_string() << "\n";
        else if (op == "c") std::cout << compare(x, y) << "\n";
    }
    return 0;
}
