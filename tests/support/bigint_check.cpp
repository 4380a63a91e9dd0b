// Test driver for relp_amd/csrc/bigint.hpp (host-only big integers used by the exact certificate).
// Reads "op a b" lines (decimal), writes the result; tests/test_bigint.py compares with Python integers.
#include <iostream>
#include <stdexcept>
#include <string>

#include "../../relp_amd/csrc/bigint.hpp"
#include "../../relp_amd/csrc/rational_reconstruct.hpp"

using relp::BigInt;

static BigInt parse(const std::string& s) {
    BigInt r(0);
    size_t pos = s[0] == '-' ? 1 : 0;
    for (; pos < s.size(); ++pos) r.mul_add_small(10, (uint32_t)(s[pos] - '0'));
    r.trim();
    if (s[0] == '-' && !r.is_zero()) r.neg = true;
    return r;
}

int main() {
    std::string op, a, b;
    while (std::cin >> op >> a >> b) {
        BigInt x = parse(a), y = parse(b);
        if (op == "add") std::cout << (x + y).to_string() << "\n";
        else if (op == "sub") std::cout << (x - y).to_string() << "\n";
        else if (op == "mul") std::cout << (x * y).to_string() << "\n";
        else if (op == "div") std::cout << (x / y).to_string() << "\n";
        else if (op == "mod") std::cout << (x % y).to_string() << "\n";
        else if (op == "gcd") std::cout << BigInt::gcd(x, y).to_string() << "\n";
        else if (op == "cmp") std::cout << cmp(x, y) << "\n";
        else if (op == "rr" || op == "rrplain" || op == "rrprime") {  // rational reconstruction of x (mod y): "n/d" or "none"
            BigInt n, d;                                               // (rrprime: y is a power of 2^31 - 1)
            if (relp::rational_reconstruct(x, y, n, d, op != "rrplain", op == "rrprime" ? 2147483647u : 0u)) std::cout << n.to_string() << "/" << d.to_string() << "\n";
            else std::cout << "none\n";
        }
        else return 2;
    }
    return 0;
}
